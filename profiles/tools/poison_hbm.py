"""Fills most of the GPU's free memory with a bit pattern (default: float32 NaNs) and frees it again, so that the next process's
fresh allocations start from garbage instead of whatever the previous run of the same program left: a read of memory the library
never wrote then shows as a changed result.  python profiles/tools/poison_hbm.py [gib] [hex word]"""
import sys
import torch
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
word = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0x7FC12345
bufs = []
for _ in range(gib):
    t = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
    t.fill_(word if word < 2**31 else word - 2**32)
    bufs.append(t)
torch.cuda.synchronize()
print("poisoned", gib, "GiB with", hex(word))
