#!/usr/bin/env python3
"""VALU-issue ceiling of the tracking kernels' interior loops, from their disassembled instruction mix x the issue rates measured
on the chip (profiles/r02_valu_issue_rates.txt, cycles of the nominal 2.4 GHz clock per wave-instruction on one SIMD at 8 waves per
SIMD) x 1024 SIMDs.  Runs in the build container (hipcc -S, no GPU):

    python profiles/tools/valu_ceiling.py            ->  profiles/r04_valu_ceilings.json

For every kernel named below it finds the interior loop (the innermost loop that contains the 16-byte global loads of the IQ
stream), counts its instructions by issue class, and reports

    cycles_per_iteration   = sum over VALU instructions of the class's cycles per wave-instruction
    samples_per_iteration  = 64 lanes x samples a lane consumes per iteration (16-byte loads x samples per 16 bytes)
    ceiling_msamples_s     = 1024 SIMDs x samples_per_iteration / cycles_per_iteration x 2.4e9 / 1e6

It is a CEILING: it assumes every SIMD issues a VALU instruction whenever it may, at the nominal clock (the chip holds a lower
clock under load, MI355X_MICROARCH.md 'DVFS give-back'), and it ignores everything outside the interior loop (prologue, ragged
chunks, reductions).  LDS reads, scalar and memory instructions are counted but priced at zero VALU cycles."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gnss-sdr-1_amd", "csrc")
CLOCK_HZ = 2.4e9
N_SIMD = 1024

# cycles per wave-instruction at 8 waves / SIMD (profiles/r02_valu_issue_rates.txt); classes by mnemonic prefix
RATES = [
    (re.compile(r"^v_pk_"), 4.25, "packed f32"),
    (re.compile(r"^v_cvt_"), 4.35, "conversion"),
    (re.compile(r"^v_lshl_add_u32|^v_lshl_or_b32|^v_add_lshl_u32|^v_mad_|^v_mul_lo|^v_mul_hi|^v_add3"), 4.35, "3-operand integer"),
    (re.compile(r"^v_(sin|cos|rcp|rsq|sqrt|exp|log)"), 8.0, "transcendental"),
    (re.compile(r"^v_.*_f64|^v_.*_u64|^v_.*_i64"), 5.6, "64-bit"),
    (re.compile(r"^v_"), 2.8, "plain 32-bit (fma / add / mul / subrev / mov / cndmask / and ...)"),
]

# name -> (source file, kernel symbol, width of the IQ loads of the interior loop, samples per load)
KERNELS = {
    "gps_l1_3tap_f32": ("trk_kernels.hip", "_Z26trk_multicorrelator_kernelILi3ELb0ELb0ELi0ELb0ELb0ELb0EE", "dwordx4", 2),
    "galileo_5tap_f32": ("trk_kernels.hip", "_Z26trk_multicorrelator_kernelILi5ELb0ELb0ELi0ELb0ELb0ELb0EE", "dwordx4", 2),
    "gps_l1_3tap_i16": ("trk_kernels.hip", "_Z26trk_multicorrelator_kernelILi3ELb0ELb0ELi1ELb0ELb0ELb0EE", "dwordx2", 2),  # cshort: 8 bytes = 2 samples per lane and load
    "closed_loop_3tap_1024": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi3ELi1024ELi0ELb0ELb0EE", "dwordx4", 2),
    "closed_loop_3tap_512": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi3ELi512ELi0ELb0ELb0EE", "dwordx4", 2),
    "closed_loop_5tap_512": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi5ELi512ELi0ELb0ELb0EE", "dwordx4", 2),
    "closed_loop_5tap_512_pilot": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi5ELi512ELi0ELb1ELb0EE", "dwordx4", 2),
    "closed_loop_5tap_1024": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi5ELi1024ELi0ELb0ELb0EE", "dwordx4", 2),
    "closed_loop_5tap_1024_pilot": ("trk_closed_loop.hip", "_Z22trk_closed_loop_kernelILi5ELi1024ELi0ELb1ELb0EE", "dwordx4", 2),
}


def assembly(src, cache):
    out = os.path.join(cache, os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(CSRC, src)):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
            "-I" + CSRC, "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernel_body(lines, sym):
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or l.startswith(sym + "v") and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].lstrip().startswith(".amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(body):
    """(first, last) line index pairs of backward branches: label ... s_cbranch* label"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.match(r"^\s+s_branch\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def analyse(body, width, samples_per_load):
    # candidates: loops whose body holds >= 2 of the 16-byte IQ loads (the interior loop keeps two in flight) and no other such loop
    # inside; the kernel has two of them -- the windowed form (LDS code window, what every bench workload runs) and the whole-table
    # form with a modulo per sample -- and the windowed one is the one with fewer VALU instructions per load
    all_loops = loops(body)
    def n16_of(a, b):
        return sum(1 for l in body[a:b + 1] if re.match(r"^\s+global_load_" + width + r"\b", l))
    cands = []
    for a, b in all_loops:
        n16 = n16_of(a, b)
        if n16 < 2:
            continue
        if any((a2 > a or b2 < b) and a2 >= a and b2 <= b and n16_of(a2, b2) >= 2 for a2, b2 in all_loops if (a2, b2) != (a, b)):
            continue
        nv = sum(1 for l in body[a:b + 1] if re.match(r"^\s+v_", l))
        cands.append((nv / n16, a, b, n16))
    if not cands:
        raise RuntimeError("no interior loop with two 16-byte global loads")
    cands.sort()
    _, a, b, n16 = cands[0]
    seg = body[a:b + 1]
    mix, cycles, n_valu, n_lds, n_salu, n_vmem = {}, 0.0, 0, 0, 0, 0
    for l in seg:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            for rx, cyc, name in RATES:
                if rx.match(op):
                    mix[name] = mix.get(name, 0) + 1
                    cycles += cyc
                    break
            n_valu += 1
        elif op.startswith("ds_"):
            n_lds += 1
        elif op.startswith("s_"):
            n_salu += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
            n_vmem += 1
    spi = 64 * n16 * samples_per_load
    return {"loop_lines": [a, b], "iq_loads_per_iteration": n16, "iq_load_width": width, "samples_per_iteration_per_wave": spi, "valu_instructions": n_valu, "lds_instructions": n_lds,
        "scalar_instructions": n_salu, "vmem_instructions": n_vmem, "valu_instructions_per_sample": n_valu * 64.0 / spi, "mix": mix,
        "cycles_per_iteration": cycles, "ceiling_msamples_s": N_SIMD * spi / cycles * CLOCK_HZ / 1e6}


def main():
    cache = os.environ.get("VALU_CEILING_CACHE", "/tmp/valu_ceiling")
    os.makedirs(cache, exist_ok=True)
    res = {"clock_hz": CLOCK_HZ, "simds": N_SIMD, "rates_source": "profiles/r02_valu_issue_rates.txt (8 waves per SIMD)",
        "rates_cycles_per_wave_instruction": {name: cyc for _, cyc, name in RATES}, "kernels": {}}
    asm = {}
    for key, (src, sym, width, spl) in KERNELS.items():
        if src not in asm:
            asm[src] = assembly(src, cache)
        body = kernel_body(asm[src], sym)
        r = analyse(body, width, spl)
        r["symbol"] = sym
        res["kernels"][key] = r
        print("%-28s %3d VALU / iteration (%.1f per sample), %.0f cycles -> ceiling %.0f Msamples/s  %s" % (key, r["valu_instructions"], r["valu_instructions_per_sample"],
            r["cycles_per_iteration"], r["ceiling_msamples_s"], r["mix"]))
    json.dump(res, open(os.path.join(ROOT, "profiles", "r04_valu_ceilings.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
