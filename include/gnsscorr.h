/*
 * gnsscorr.h -- C ABI of libgnsscorr.so, the MI355X (gfx950, HIP) acquisition +
 * tracking correlator engine.
 *
 * This is the drop-in boundary for ONE path of GNSS-SDR (zhufengGNSS/gnss-sdr-1):
 *
 *   tracking   Cpu_Multicorrelator_Real_Codes                      (reference:
 *              src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:45-69)
 *              = volk_gnsssdr_32f_xn_resampler_32f_xn               (code NCO)
 *              + volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn     (carrier NCO + E/P/L)
 *   acquisition pcps_acquisition                                    (reference:
 *              src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.h:81-261,
 *              .cc:239-274 set_local_code, :313-368 init, :668-927 acquisition_core)
 *
 * Conventions: plain C types only (no C++/torch types); every function returns
 * a gc_status (0 = ok) and records a message readable with gc_last_error();
 * no exception crosses the ABI.  "host" pointers are caller-owned host memory,
 * "dev" pointers are caller-owned device (HBM) memory on the context's GPU.
 * Complex values are interleaved float32 (re, im), i.e. std::complex<float> /
 * gr_complex / lv_32fc_t.  A `stream` argument is a hipStream_t passed as
 * void* (NULL = the context's own stream).  The library fails loudly
 * (GC_ERR_NO_DEVICE) when no HIP device is usable: there is no CPU fallback.
 *
 * Threading follows the reference (one correlator / acquisition object per
 * channel, driven by that channel's thread): a handle is used by one thread at
 * a time; different handles of one context may be used from different threads
 * at once (their calls are serialised on the context's mutex and stream).
 *
 * The C++ classes in gnss-sdr-1_amd/adapter/ (Hip_Multicorrelator_Real_Codes,
 * hip_pcps_acquisition, the TrackingInterface/AcquisitionInterface-shaped
 * adapters) are thin inline wrappers over these entry points; INTEGRATION.md
 * shows the binding a GNSS-SDR maintainer adds.
 */
#ifndef GNSSCORR_H
#define GNSSCORR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int gc_status;
enum
{
    GC_OK = 0,
    GC_ERR_INVALID = 1,   /* bad argument / shape */
    GC_ERR_NO_DEVICE = 2, /* no usable HIP device (no CPU fallback exists) */
    GC_ERR_HIP = 3,       /* a HIP runtime call failed */
    GC_ERR_STATE = 4      /* call sequence error (e.g. correlate before init) */
};

/* Last error message of the calling thread ("" when none). */
const char* gc_last_error(void);
/* Library version string. */
const char* gc_version(void);
/* Layout check between a binding and the loaded library: the sizes of the structures that cross the ABI, as the CALLER's
 * header (or ctypes / cgo / JNI mirror) sees them.  GC_OK when they all match the library's.  C and C++ callers use
 * GC_ABI_CHECK() after including this header (the macro is at its end). */
gc_status gc_abi_check(size_t sizeof_epoch_params, size_t sizeof_loop_conf, size_t sizeof_loop_record, size_t sizeof_loop_sync_conf,
    size_t sizeof_acq_conf, size_t sizeof_acq_result);
/* Number of visible HIP devices (0 when none; never fails). */
int gc_device_count(void);
/* 1 when the library was built with -DGNSSCORR_EXPERIMENTS: it then also carries the measured-slower kernel variants and reads
 * their GNSSCORR_* tuning variables.  The product build returns 0 and reads none of them. */
int gc_build_has_experiments(void);

/* ------------------------------------------------------------------------ */
/* Context: one per GPU.  Owns a HIP stream and scratch buffers.             */
/* ------------------------------------------------------------------------ */
typedef struct gc_ctx gc_ctx;
gc_status gc_ctx_create(int device, gc_ctx** out);
/* Drops the caller's reference.  Handles created on the context keep it (and its
 * stream) alive until the last of them is destroyed, so the order of the destroy
 * calls does not matter; the same holds for a gc_stream and the batches reading it. */
gc_status gc_ctx_destroy(gc_ctx* ctx);
gc_status gc_ctx_synchronize(gc_ctx* ctx);

/* ------------------------------------------------------------------------ */
/* Level 1 -- one correlator object per channel, host pointers, synchronous. */
/* Method-for-method image of Cpu_Multicorrelator_Real_Codes                 */
/* (cpu_multicorrelator_real_codes.h:48-57).  Pointer arguments are RETAINED, */
/* not copied, exactly like the reference (cpu_multicorrelator_real_codes.cc: */
/* 79-98): `shifts_chips` and the code table are re-read on every correlate   */
/* call, so the caller may edit the shifts between calls.                     */
/* ------------------------------------------------------------------------ */
typedef struct gc_correlator gc_correlator;
gc_status gc_correlator_create(gc_ctx* ctx, gc_correlator** out);
gc_status gc_correlator_destroy(gc_correlator* c);
/* ::set_high_dynamics_resampler (.cc:189-193).  The reference constructor
 * defaults this flag to TRUE (.cc:49); so does gc_correlator_create. */
gc_status gc_correlator_set_high_dynamics_resampler(gc_correlator* c, int use_high_dynamics_resampler);
/* ::init (.cc:62-76) */
gc_status gc_correlator_init(gc_correlator* c, int max_signal_length_samples, int n_correlators);
/* ::set_local_code_and_taps (.cc:79-89) */
gc_status gc_correlator_set_local_code_and_taps(gc_correlator* c, int code_length_chips,
    const float* local_code_in, float* shifts_chips);
/* ::set_input_output_vectors (.cc:92-98); corr_out: n_correlators complex,
 * sig_in: >= signal_length_samples complex (host memory) */
gc_status gc_correlator_set_input_output_vectors(gc_correlator* c, float* corr_out, const float* sig_in);
/* ::Carrier_wipeoff_multicorrelator_resampler, 7-argument form (.cc:129-152) */
gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples);
/* 6-argument form (.cc:155-170): always the plain rotator; the resampler still
 * follows the high-dynamics flag */
gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler_6(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples);
/* ::free (.cc:173-186) */
gc_status gc_correlator_free(gc_correlator* c);
/* Engine statistics (no reference counterpart).  Concurrent calls of Carrier_wipeoff_multicorrelator_resampler from the
 * channel threads of one context are combined into batches, one kernel launch each ("epoch batcher"): launches so far, calls
 * served, calls whose input window was shared with another call of the same batch (same sig_in pointer: one copy to the GPU
 * for the group), and the largest batch.  Any pointer may be NULL. */
gc_status gc_correlator_batch_stats(gc_ctx* ctx, uint64_t* n_batches, uint64_t* n_requests, uint64_t* n_shared_windows, int* max_batch);
/* Optional, no reference counterpart: page-locks caller memory that the correlators of this context read their input from --
 * typically the GNU Radio buffer behind the tracking blocks' input port (every channel's sig_in points into it,
 * gnss_flowgraph.cc:496-499).  Windows inside a registered buffer go to the GPU without the staging copy, and the windows of one
 * batch that overlap (channels at neighbouring read positions) cross PCIe once, as one transfer of their union.  The memory must
 * stay valid until it is unregistered or the context is destroyed.  Fails (and changes nothing) where the runtime cannot pin the
 * range, e.g. some doubly mapped circular buffers. */
gc_status gc_ctx_register_host_buffer(gc_ctx* ctx, const void* base, size_t bytes);
gc_status gc_ctx_unregister_host_buffer(gc_ctx* ctx, const void* base);

/* The same object with COMPLEX chips is the image of Cpu_Multicorrelator
 * (cpu_multicorrelator.h:46-64; GLONASS L1/L2 and the GPS L1 C-Aid trackers):
 * volk_gnsssdr_32fc_xn_resampler_32fc_xn (.cc:103-113) followed by
 * volk_gnsssdr_32fc_x2_rotator_dot_prod_32fc_xn (.cc:116-130).  It has no
 * carrier-rate / code-rate arguments and no high-dynamics variant.
 * local_code_in_iq: code_length_chips (re, im) pairs, pointer retained. */
gc_status gc_correlator_set_local_code_and_taps_complex(gc_correlator* c, int code_length_chips,
    const float* local_code_in_iq, float* shifts_chips);
/* Cpu_Multicorrelator::Carrier_wipeoff_multicorrelator_resampler, 5 arguments
 * (cpu_multicorrelator.cc:116-130); GC_ERR_STATE unless the code is complex
 * (float pairs, or lv_16sc_t with the setters below). */
gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips,
    int signal_length_samples);

/* With lv_16sc_t chips, input and output the object is the image of
 * Cpu_Multicorrelator_16sc (cpu_multicorrelator_16sc.h:44-67; the *_sc C-Aid
 * trackers): volk_gnsssdr_16ic_xn_resampler_16ic_xn followed by
 * volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn (.cc:78-103), run with the
 * 5-argument call above.  Arithmetic: each rotated sample is rounded to int16
 * (rintf), multiplied with the int16 chip (int16 wrap) and summed.  The sums
 * are formed exactly in 32 bits and saturated once, which equals the
 * reference's running saturating sum whenever no intermediate sum leaves the
 * int16 range; the rounding of a sample can differ by one LSB where the
 * reference's sequentially rounded float phase puts it on the other side of
 * x.5.  int16 arrays are (re, im) interleaved. */
gc_status gc_correlator_set_local_code_and_taps_16sc(gc_correlator* c, int code_length_chips,
    const int16_t* local_code_in_iq, float* shifts_chips);
gc_status gc_correlator_set_input_output_vectors_16sc(gc_correlator* c, int16_t* corr_out, const int16_t* sig_in);

/* ------------------------------------------------------------------------ */
/* RF stream ring: the IQ samples of one RF stream, pushed to a GPU once and    */
/* read by every channel / acquisition of that stream (all channels of a        */
/* GNSS-SDR flowgraph read the same stream, gnss_flowgraph.cc:496-499).  A ring  */
/* of capacity_samples in HBM plus a mirror of its first max_window_samples, so  */
/* that any window of <= max_window_samples is contiguous; consumers address it  */
/* with ABSOLUTE sample numbers (0 = first sample ever pushed).  Pushes are      */
/* asynchronous (own HIP stream, pinned staging) and overlap with compute; a     */
/* push waits only for launches that may still read the samples it evicts.       */
/* ------------------------------------------------------------------------ */
typedef struct gc_stream gc_stream;
gc_status gc_stream_create(gc_ctx* ctx, int iq_format, uint64_t capacity_samples, uint32_t max_window_samples,
    gc_stream** out);
gc_status gc_stream_destroy(gc_stream* s);
/* Appends n_samples (host memory, gc_iq_format of the stream; at most capacity_samples per call).
 * first_index (optional) receives the absolute number of the first appended sample. */
gc_status gc_stream_push(gc_stream* s, const void* host_iq, uint64_t n_samples, uint64_t* first_index);
/* Same for a PAGE-LOCKED host buffer (hipHostMalloc / hipHostRegister, e.g. a pinned torch tensor): the DMA
 * reads it directly, without the staging copy, so it must stay untouched until gc_stream_synchronize(). */
gc_status gc_stream_push_pinned(gc_stream* s, const void* pinned_host_iq, uint64_t n_samples, uint64_t* first_index);
/* The same page-locked block into several rings -- the RF stream's copy on each GPU of a node (channels are
 * sharded over the GPUs, every GPU needs the whole stream, no collective): G independent H2D copies enqueued back to back, each
 * on its ring's own copy stream.  The rings share the sample format; the block stays untouched until every ring is synchronised. */
gc_status gc_stream_broadcast_pinned(gc_stream* const* rings, int n_rings, const void* pinned_host_iq, uint64_t n_samples);
/* Resident range [oldest_index, head_index) and the ring capacity (any pointer may be NULL). */
gc_status gc_stream_info(gc_stream* s, uint64_t* oldest_index, uint64_t* head_index, uint64_t* capacity_samples);
/* Waits until every push so far has landed in HBM. */
gc_status gc_stream_synchronize(gc_stream* s);

/* ------------------------------------------------------------------------ */
/* Level 2 -- batched tracking engine: all channels of a GPU, many epochs,    */
/* one launch; IQ, parameters and results resident in HBM.                    */
/* ------------------------------------------------------------------------ */

/* One channel-epoch of work: the arguments the reference hands to its two
 * kernels (resampler_32f_xn.h:77 / rotator_dot_prod_32fc_xn.h:81) for one call
 * of Carrier_wipeoff_multicorrelator_resampler, plus where the window starts. */
typedef struct
{
    uint64_t sample_offset;    /* first IQ sample of the window, relative to the channel's IQ base */
    float phase0_re, phase0_im;        /* lv_cmake(cos(rem_carr), -sin(rem_carr))   (.cc:141) */
    float phase_inc_re, phase_inc_im;  /* std::exp(lv_32fc_t(0, -phase_step_rad))   (.cc:149) */
    float phase_rate_re, phase_rate_im;/* std::exp(lv_32fc_t(0, -phase_rate_step))  (.cc:145); (1,0) = none */
    float rem_code_phase_chips;        /* in code samples = chips * samples_per_chip */
    float code_phase_step_chips;
    float code_phase_rate_step_chips;
    int32_t n_samples;                 /* integration length (signal_length_samples) */
} gc_epoch_params; /* 48 bytes */

/* Fills a gc_epoch_params from the reference's scalar arguments with the same
 * float arithmetic as cpu_multicorrelator_real_codes.cc:141-149 (host libm). */
void gc_epoch_params_fill(gc_epoch_params* p, uint64_t sample_offset,
    float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples);

/* IQ sample formats accepted from HBM.  Integer formats are what SDR front-ends deliver and the reference
 * converts to gr_complex before its float correlators (volk_gnsssdr_16ic_convert_32fc,
 * pcps_acquisition.cc:676-679; data_type_adapter blocks): the engine converts on load (plain cast), so
 * results equal the float path on the converted samples while HBM bytes per sample drop from 8 to 4 / 2. */
typedef enum
{
    GC_IQ_F32 = 0, /* interleaved float32 (re, im): gr_complex / lv_32fc_t */
    GC_IQ_I16 = 1, /* interleaved int16   (re, im): lv_16sc_t ("cshort") */
    GC_IQ_I8 = 2   /* interleaved int8    (re, im): lv_8sc_t  ("cbyte") */
} gc_iq_format;

typedef struct gc_trk_batch gc_trk_batch;
/* n_channels channels with n_taps correlator taps each; code tables up to
 * max_code_length entries.  high_dyn selects the high-dynamics resampler +
 * rotator pair for every channel of the batch. */
gc_status gc_trk_batch_create(gc_ctx* ctx, int n_channels, int n_taps, int max_code_length,
    int high_dyn, gc_trk_batch** out);
gc_status gc_trk_batch_destroy(gc_trk_batch* b);
/* Uploads channel `ch`'s code table (float[code_length], +-1 or any real
 * replica) and tap shifts (float[n_taps], in code samples). (host pointers)
 * Setters take effect at the next run call; they wait for launches on the
 * context's own stream, but launches still in flight on a CALLER stream
 * (gc_trk_batch_run_dev) must be synchronised by the caller first. */
gc_status gc_trk_batch_set_code(gc_trk_batch* b, int ch, const float* code, int code_length,
    const float* shifts_chips);
gc_status gc_trk_batch_set_shifts(gc_trk_batch* b, int ch, const float* shifts_chips);
/* Complex chips for every channel of the batch (Cpu_Multicorrelator,
 * cpu_multicorrelator.cc:82-130): switch the batch with set_complex_codes(b, 1)
 * (drops the codes loaded so far; not available with high_dyn; max_code_length
 * <= 7936), then load code_length (re, im) pairs per channel. */
gc_status gc_trk_batch_set_complex_codes(gc_trk_batch* b, int on);
gc_status gc_trk_batch_set_code_complex(gc_trk_batch* b, int ch, const float* code_iq, int code_length,
    const float* shifts_chips);
/* Cpu_Multicorrelator_16sc arithmetic for every channel of the batch (see
 * gc_correlator_set_local_code_and_taps_16sc): set_16sc(b, 1) switches the
 * input format to GC_IQ_I16 and drops codes and inputs loaded so far (not
 * available with high_dyn); codes are code_length (re16, im16) pairs; the
 * output of run / run_dev is n_taps lv_16sc_t (4 bytes) per channel-epoch. */
gc_status gc_trk_batch_set_16sc(gc_trk_batch* b, int on);
gc_status gc_trk_batch_set_code_16sc(gc_trk_batch* b, int ch, const int16_t* code_iq, int code_length,
    const float* shifts_chips);
/* Sample format of every channel's IQ buffer (default GC_IQ_F32). */
gc_status gc_trk_batch_set_input_format(gc_trk_batch* b, int iq_format);
/* Points channel `ch` at its IQ samples in HBM (n_samples complex samples of the batch's format,
 * aligned to one sample).  Channels of one RF stream may share the same pointer. */
gc_status gc_trk_batch_set_input_dev(gc_trk_batch* b, int ch, const void* dev_iq, uint64_t n_samples);
/* Channel `ch` reads the ring `s` (same format as the batch): gc_epoch_params.sample_offset is then an
 * ABSOLUTE sample number of that stream and n_samples <= the stream's max_window.  gc_trk_batch_run checks
 * every window against the resident range; gc_trk_batch_run_dev cannot (parameters live in HBM): tell it the
 * oldest sample its launches read with set_read_floor so that later pushes need not wait for them (default:
 * everything resident, i.e. the next evicting push waits for the launch). */
gc_status gc_trk_batch_set_input_stream(gc_trk_batch* b, int ch, gc_stream* s);
gc_status gc_trk_batch_set_read_floor(gc_trk_batch* b, uint64_t oldest_index_read);
/* Correlates n_epochs epochs of every channel.  dev_params: n_channels*n_epochs
 * gc_epoch_params, channel-major.  dev_out: n_channels*n_epochs*n_taps complex.
 * Asynchronous on `stream`. */
gc_status gc_trk_batch_run_dev(gc_trk_batch* b, int n_epochs, const gc_epoch_params* dev_params,
    void* dev_out, void* stream);
/* Same with host parameter/result buffers (copies + synchronises). */
gc_status gc_trk_batch_run(gc_trk_batch* b, int n_epochs, const gc_epoch_params* host_params,
    float* host_out);
/* Engine tuning (no reference counterpart).  Nominal integration length: lets
 * gc_trk_batch_run_dev pick how many slices to cut an epoch into when the
 * batch alone would not fill the GPU, and size the LDS code window of a launch by what
 * one slice of such an epoch touches (long codes: Galileo E1's 8184 samples are cut in
 * two so that a workgroup holds half the table; gc_trk_batch_run_dev assumes ONE code period
 * per nominal window, gc_trk_batch_run reads the code steps of its records).  Results never
 * depend on it: a record outside the bound is served from the whole table, in LDS when the
 * launch's LDS holds it, from global memory otherwise (slower: a batch of long codes whose
 * windows span several code periods should use set_slices(-1)).  set_slices(0) = automatic;
 * set_slices(-1) = automatic slicing by load only, the window sized for the longest code
 * (the behaviour before round 4; A/B timing). */
gc_status gc_trk_batch_set_nominal_length(gc_trk_batch* b, int n_samples);
gc_status gc_trk_batch_set_slices(gc_trk_batch* b, int n_slices);

/* ------------------------------------------------------------------------ */
/* Level 3 -- closed-loop tracking on the device: the correlations AND the      */
/* per-epoch DLL/PLL maths of dll_pll_veml_tracking run inside one launch for    */
/* n_epochs code periods per channel (no host round trip per millisecond).       */
/* ------------------------------------------------------------------------ */

/* What the tracking block knows when start_tracking() is called: the Dll_Pll_Conf fields it reads
 * (src/algorithms/tracking/libs/dll_pll_conf.h:39-80), the per-signal constants of its constructor
 * (dll_pll_veml_tracking.cc:113-330) and the acquisition hand-over from Gnss_Synchro (:555-557). */
typedef struct
{
    double fs_in;
    double signal_carrier_freq_hz;
    double code_chip_rate_hz;
    double code_period_s;
    double carrier_lock_th;
    double acq_delay_samples;          /* Gnss_Synchro::Acq_delay_samples */
    double acq_doppler_hz;             /* Gnss_Synchro::Acq_doppler_hz */
    uint64_t acq_samplestamp_samples;  /* Gnss_Synchro::Acq_samplestamp_samples */
    uint64_t sample_counter;           /* stream sample count at the first sample of the channel's IQ buffer */
    uint32_t code_length_chips;
    uint32_t code_samples_per_chip;
    uint32_t vector_length;
    uint32_t pull_in_time_s;
    int32_t veml;                      /* != 0: 5 taps VE/E/P/L/VL (Galileo E1), else 3 taps E/P/L */
    int32_t pll_filter_order, dll_filter_order;
    int32_t enable_fll_pull_in, enable_fll_steady_state;
    int32_t cn0_samples, cn0_min, max_lock_fail;
    float pll_bw_hz, dll_bw_hz, fll_bw_hz;
    float early_late_space_chips, very_early_late_space_chips;
    uint32_t high_dyn_smoother_length; /* 0: Dll_Pll_Conf::high_dyn false; n > 0: high_dyn with smoother_length n (<= 16): the
                                        * high-dynamics resampler / rotator kernels and the carrier / code rate smoothers of
                                        * update_tracking_vars (:1016-1033, :1047-1064) */
} gc_loop_conf;

/* One code period of one channel: the correlator outputs plus what the block writes to Gnss_Synchro
 * (dll_pll_veml_tracking.cc:1730-1770, 1898-1906) and to its binary dump (:1196-1243). */
typedef struct
{
    float corr[10];              /* n_taps complex correlator outputs (re, im) */
    float carrier_doppler_hz, code_freq_chips;
    float carr_phase_error_hz, carr_error_filt_hz, code_error_chips, code_error_filt_chips;
    float cn0_db_hz, carrier_lock_test;
    uint64_t sample_counter;     /* Tracking_sample_counter after this epoch */
    double acc_carrier_phase_rad;
    double rem_code_phase_samples;
    int32_t state;               /* d_state after this code period: 0 = standby (after loss of lock), 1 = pull-in pending,
                                  * 2 = wide tracking / symbol synchronisation, 3 = extended integration, 4 = narrow tracking */
    int32_t valid;               /* Flag_valid_symbol_output */
    int32_t current_prn_length_samples;
    int32_t extend_count;        /* d_extend_correlation_symbols_count when log_data ran (scale factor of the dump, :1179-1191) */
    float accu[10];              /* d_VE/E/P/L/VL_accu as log_data sees them (before the reset that follows a loop update) */
    float prompt_data[2];        /* d_Prompt_Data (pilot tracking: prompt of the data component), else the prompt tap */
    int32_t integrating;         /* 1: the period only accumulated (state 3, log_data(true)); 0: the loop filters ran */
    int32_t reserved;
} gc_loop_record;

/* Symbol synchronisation, extended integration and pilot tracking (dll_pll_veml_tracking.cc:1601-1896): what the block's
 * constructor derives from the signal (:113-336) plus the Dll_Pll_Conf fields of the narrow stage.  Without it a channel
 * stays in state 2 (1 code period per loop update, data component), which is what the block does for a signal whose
 * telemetry preamble is never found. */
typedef struct
{
    int32_t extend_correlation_symbols;   /* Dll_Pll_Conf::extend_correlation_symbols (>= 1; 1 = no extension) */
    int32_t track_pilot;                  /* Dll_Pll_Conf::track_pilot: `code` of gc_trk_loop_start is the pilot replica,
                                           * data_code the data component's; four-quadrant PLL once the secondary code is locked */
    int32_t symbols_per_bit;              /* d_symbols_per_bit */
    int32_t secondary_code_length;        /* d_secondary_code_length, 0 = no secondary code (<= 128) */
    int32_t preamble_length_symbols;      /* d_preamble_length_symbols, 0 = none (<= 192) */
    float bit_sync_min_time_s;            /* tracking time before the preamble search starts: 10 in the reference (:1648) */
    float pll_bw_narrow_hz, dll_bw_narrow_hz;
    float early_late_space_narrow_chips, very_early_late_space_narrow_chips;
    char secondary_code[128];             /* '0' / '1', like the reference's secondary-code strings */
    int8_t preamble_symbols[192];         /* +1 / -1 (d_preambles_symbols) */
} gc_loop_sync_conf;

/* Fills the signal-dependent part of a gc_loop_sync_conf the way the block's constructor and start_tracking do
 * (dll_pll_veml_tracking.cc:113-336, :631-705): symbols per bit, secondary code, telemetry preamble for
 * system / signal 'G' "1C" | "2S" | "L5", 'E' "1B" | "5X", 'C' "B1" | "B3" and the satellite `prn` (BeiDou GEO 1..5 broadcast D2;
 * Galileo E5a-Q has one secondary code per PRN).  track_pilot is cleared for signals without a pilot component;
 * bit_sync_min_time_s is set to the reference's 10 s; the narrow-stage fields are left for the caller (Dll_Pll_Conf). */
gc_status gc_loop_sync_for_signal(char system, const char* signal, uint32_t prn, int track_pilot, int extend_correlation_symbols,
    gc_loop_sync_conf* out);

typedef struct gc_trk_loop gc_trk_loop;
gc_status gc_trk_loop_create(gc_ctx* ctx, int n_channels, int max_code_length, gc_trk_loop** out);
gc_status gc_trk_loop_destroy(gc_trk_loop* l);
/* Sample format of every channel's input (default GC_IQ_F32; cshort / cbyte samples are converted on load like
 * in the batched engine).  Call before binding inputs. */
gc_status gc_trk_loop_set_input_format(gc_trk_loop* l, int iq_format);
/* IQ block of channel `ch` in HBM (samples of the engine's format); epochs are correlated until it is exhausted. */
gc_status gc_trk_loop_set_input_dev(gc_trk_loop* l, int ch, const void* dev_iq, uint64_t n_samples);
/* Channel `ch` reads the RF stream ring `s` (same sample format) instead of a fixed block: bind before
 * gc_trk_loop_start; the channel then starts at absolute sample gc_loop_conf.sample_counter and every launch
 * correlates the code periods that are complete in the ring at that moment (the remaining records of the
 * launch are marked invalid, state unchanged), so "push a block, run" is the whole host loop.
 * gc_trk_loop_run reports GC_ERR_STATE when a channel has fallen behind the ring's oldest sample. */
gc_status gc_trk_loop_set_input_stream(gc_trk_loop* l, int ch, gc_stream* s);
/* Installs (sync != NULL) or removes the synchronisation / extension description of channel `ch`; it takes effect at the
 * next gc_trk_loop_start.  data_code (data_code_length floats, same length as the tracking replica) is required with
 * track_pilot and ignored otherwise.  All channels of one engine share the pilot mode. */
gc_status gc_trk_loop_set_sync(gc_trk_loop* l, int ch, const gc_loop_sync_conf* sync, const float* data_code, int data_code_length);
/* dll_pll_veml_tracking::start_tracking (:549-747): uploads the replica (code_length_chips *
 * code_samples_per_chip floats), sets the taps from the spacings and initialises the loop. */
gc_status gc_trk_loop_start(gc_trk_loop* l, int ch, const gc_loop_conf* conf, const float* code, int code_length);
/* Puts channel `ch` back in standby (all-zero records, state 0) until the next gc_trk_loop_start; channels that were never
 * started are in the same state, so an engine can be sized for the receiver's channel count and filled as acquisitions succeed. */
gc_status gc_trk_loop_stop(gc_trk_loop* l, int ch);
/* n_epochs code periods of every channel in ONE launch.  dev_records: n_channels*n_epochs records,
 * channel-major.  The loop state persists on the device between calls. */
/* Launch geometry of the engine, for tests and tuning (0 = automatic, the default): threads per workgroup (256, 512 or 1024; the
 * default follows the channel count) and workgroups per channel-period.  slices_per_channel > 1 (a period cut into slices, one launch
 * per code period, the last slice to finish runs the loop maths) measured slower than one workgroup per channel (13.4 vs 11.4 us per
 * period at 32 channels) and is accepted by experiments builds only; the product library takes 0 or 1. */
gc_status gc_trk_loop_set_geometry(gc_trk_loop* l, int threads_per_workgroup, int slices_per_channel);
gc_status gc_trk_loop_run_dev(gc_trk_loop* l, int n_epochs, gc_loop_record* dev_records, void* stream);
gc_status gc_trk_loop_run(gc_trk_loop* l, int n_epochs, gc_loop_record* host_records);

/* ------------------------------------------------------------------------ */
/* PRN replica generators (host side, set-up path).  Same outputs as the      */
/* reference's generators; dest buffers are caller-owned host memory.         */
/* ------------------------------------------------------------------------ */
/* gps_l1_ca_code_gen_float (src/algorithms/libs/gps_sdr_signal_processing.cc:119-130): 1023 chips, +-1;
 * PRN 1..32 and SBAS 120..138 */
gc_status gc_gps_l1_ca_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift);
/* gps_l1_ca_code_gen_complex_sampled (…:151-196): (int)(fs/1000) complex samples; *n_samples (optional) = count */
gc_status gc_gps_l1_ca_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples);
/* glonass_l1_ca_code_gen_complex / _complex_sampled (src/algorithms/libs/glonass_l1_signal_processing.cc:37-153;
 * glonass_l2_signal_processing.cc is the same sequence): 511 chips, one code for every satellite (FDMA).  The float
 * form holds the real parts (the reference's chips are (+-1, 0)); sampled: (int)(fs/1000) complex samples. */
gc_status gc_glonass_l1_ca_code_gen_float(float* dest, uint32_t chip_shift);
gc_status gc_glonass_l1_ca_code_gen_complex_sampled(float* dest, int32_t fs, uint32_t chip_shift, int32_t* n_samples);
/* beidou_b1i_code_gen_float / _complex_sampled (src/algorithms/libs/beidou_b1i_signal_processing.cc:115-191): 2046 chips */
gc_status gc_beidou_b1i_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift);
gc_status gc_beidou_b1i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples);
/* galileo_e1_code_gen_sinboc11_float (src/algorithms/libs/galileo_e1_signal_processing.cc:108-119): 8184 samples
 * (2 per chip) of the E1-B ("1B") or E1-C ("1C") primary code, PRN 1..50.  The memory codes are read from
 * data/galileo_e1_primary_codes.bin next to the library (or $GNSSCORR_GALILEO_E1_CODES). */
gc_status gc_galileo_e1_code_gen_sinboc11_float(float* dest, const char* signal, uint32_t prn);
/* galileo_e1_code_gen_complex_sampled (…:232-255) without secondary code: 4 ms of samples at fs */
gc_status gc_galileo_e1_code_gen_complex_sampled(float* dest, const char* signal, int32_t cboc, uint32_t prn, int32_t fs,
    uint32_t chip_shift, int32_t* n_samples);
/* The 10.23 / 0.5115 Mcps signals.  Their per-PRN constants (IS-GPS-200 Table 3-IIa, IS-GPS-705 Table 3-Ia/Ib, BDS-SIS-ICD-B3I
 * G2 phases, Galileo OS SIS ICD E5a memory codes) are read from data/prn_tables.bin and data/galileo_e5a_primary_codes.bin
 * next to the library (or $GNSSCORR_DATA_DIR).  All codes are 10230 chips.
 * gps_l2c_m_code_gen_float / _complex_sampled (src/algorithms/libs/gps_l2c_signal.cc:75-137): PRN 1..50; the sampled form
 * holds (int)(fs / 50) complex samples (one 20 ms code) and digitises with a true ceil() */
gc_status gc_gps_l2c_m_code_gen_float(float* dest, uint32_t prn);
gc_status gc_gps_l2c_m_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples);
/* gps_l5i / gps_l5q _code_gen_float / _complex_sampled (src/algorithms/libs/gps_l5_signal.cc:197-344): PRN 1..50 */
gc_status gc_gps_l5i_code_gen_float(float* dest, uint32_t prn);
gc_status gc_gps_l5q_code_gen_float(float* dest, uint32_t prn);
gc_status gc_gps_l5i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples);
gc_status gc_gps_l5q_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples);
/* beidou_b3i_code_gen_float / _complex_sampled (src/algorithms/libs/beidou_b3i_signal_processing.cc:37-246): PRN 1..63 */
gc_status gc_beidou_b3i_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift);
gc_status gc_beidou_b3i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples);
/* galileo_e5_a_code_gen_complex_primary / _complex_sampled (src/algorithms/libs/galileo_e5_signal_processing.cc:38-142):
 * 10230 complex chips; signal "5I" -> (I, 0), "5Q" -> (0, Q), "5X" -> (I, Q); PRN 1..50 */
gc_status gc_galileo_e5_a_code_gen_complex_primary(float* dest, int32_t prn, const char* signal);
gc_status gc_galileo_e5_a_code_gen_complex_sampled(float* dest, const char* signal, uint32_t prn, int32_t fs, uint32_t chip_shift,
    int32_t* n_samples);
/* Secondary (overlay) code of a signal as a '0'/'1' string, the form gc_loop_sync_conf.secondary_code takes and the reference
 * keeps in its system_parameters headers: "1C" (Galileo E1-C CS25), "B1" / "B3" (BeiDou NH20), "L5I" (NH10), "L5Q" (NH20),
 * "5I" (Galileo E5a-I CS20), "5Q" (E5a-Q CS100 of `prn`; the reference's table holds PRN 1..47).  prn is ignored otherwise. */
gc_status gc_secondary_code(const char* signal, uint32_t prn, char* dest, int32_t capacity, int32_t* length);

/* ------------------------------------------------------------------------ */
/* Acquisition -- PCPS (parallel code phase search), batched over satellites. */
/* ------------------------------------------------------------------------ */

/* Image of Acq_Conf (src/algorithms/acquisition/libs/acq_conf.h:38-66), the
 * fields pcps_acquisition reads on this path, plus doppler_step (set through
 * set_doppler_step in the reference, pcps_acquisition.h:228). */
typedef struct
{
    int64_t fs_in;
    uint32_t sampled_ms;
    uint32_t ms_per_code;
    float samples_per_ms;
    float samples_per_code;
    uint32_t samples_per_chip;
    uint32_t doppler_max;
    uint32_t doppler_step;
    uint32_t max_dwells;
    int32_t bit_transition_flag;
    int32_t use_CFAR_algorithm_flag;
    /* engine extension: when > 0 overrides ceil(2*doppler_max/doppler_step)
     * (pcps_acquisition.cc:326) as the number of Doppler bins */
    uint32_t num_doppler_bins_override;
    /* two-step acquisition (Acq_Conf::make_2_steps, num_doppler_bins_step2, doppler_step2;
     * defaults 4 bins of 125 Hz, gps_l1_ca_pcps_acquisition.cc:86-88) */
    int32_t make_2_steps;
    uint32_t num_doppler_bins_step2;
    float doppler_step2;
} gc_acq_conf;

/* Per-satellite result of one dwell: what acquisition_core leaves in
 * Gnss_Synchro / d_test_statistics / d_mag / d_input_power (:747-768). */
typedef struct
{
    uint32_t indext;          /* argmax code-phase index in the grid row */
    int32_t doppler_hz;       /* -doppler_max + doppler_step*row (:588) */
    uint32_t doppler_index;
    float test_statistics;    /* CFAR: max/N^4/input_power; else first/second peak */
    float mag;                /* raw grid maximum */
    float input_power;
    float second_peak;          /* as the reference computes it (the N-byte memcpy at :647) */
    float second_peak_full_row; /* second peak with the whole peak row considered */
    double acq_delay_samples; /* fmod((float)indext, samples_per_code) (:766) */
    double acq_doppler_hz;
} gc_acq_result;

typedef struct gc_acq gc_acq;
/* pcps_acquisition ctor + init() for n_sats satellites searched at once. */
gc_status gc_acq_create(gc_ctx* ctx, const gc_acq_conf* conf, int n_sats, gc_acq** out);
gc_status gc_acq_destroy(gc_acq* a);
gc_status gc_acq_fft_size(const gc_acq* a, uint32_t* fft_size, uint32_t* consumed_samples,
    uint32_t* num_doppler_bins);
/* pcps_acquisition::set_local_code for satellite slot `sat` (host pointer to
 * consumed_samples complex; fft_size/2 with bit_transition_flag). */
gc_status gc_acq_set_local_code(gc_acq* a, int sat, const float* code);
/* d_old_freq of the reference (pcps_acquisition.cc:242-247, :276-293, :371-380): a frequency added to every bin of
 * the coarse Doppler grid -- an intermediate frequency, or the GLONASS FDMA channel offset DFRQ1_GLO * k the reference
 * installs in set_local_code() for "1G" / "2G" signals.  Rebuilds the wipe-off table (float32 running phase);
 * reported Doppler values stay relative to the offset, as in the reference.  Default 0. */
gc_status gc_acq_set_frequency_offset(gc_acq* a, int64_t offset_hz);
/* New search: the dwell counter restarts and the magnitude grids read as zero (the first dwell after a
 * reset overwrites them; nothing is enqueued by this call, so it needs no stream). */
gc_status gc_acq_reset(gc_acq* a);
/* Second step of make_2_steps (pcps_acquisition.cc:771-829, 957-963): enable != 0 switches the search to
 * num_doppler_bins_step2 bins of doppler_step2 Hz centred on doppler_center_hz
 * (update_grid_doppler_wipeoffs_step2, :383-390) and restarts the dwell counter; enable == 0 returns to
 * the coarse grid.  Results then follow the step-two Doppler formula (:589-591). */
gc_status gc_acq_set_step_two(gc_acq* a, int enable, float doppler_center_hz);
/* Sample format of the device input blocks of gc_acq_dwell_dev / gc_acq_dwell_enqueue (default GC_IQ_F32;
 * GC_IQ_I16 is the block's "cshort" item type, converted like volk_gnsssdr_16ic_convert_32fc does,
 * pcps_acquisition.cc:676-679).  gc_acq_dwell (host pointer) always takes gr_complex. */
gc_status gc_acq_set_input_format(gc_acq* a, int iq_format);
/* One dwell of acquisition_core for every satellite on the same input block
 * (consumed_samples complex).  Non-coherent accumulation across calls like the
 * reference (d_num_noncoherent_integrations_counter).  results: n_sats. */
gc_status gc_acq_dwell_dev(gc_acq* a, const void* dev_iq, gc_acq_result* host_results, void* stream);
gc_status gc_acq_dwell(gc_acq* a, const float* host_iq, gc_acq_result* host_results);
/* Enqueue-only variant for throughput runs: results stay in HBM until
 * gc_acq_fetch_results().  The input block has been consumed (in stream order) when the call returns its work to `stream`.
 * Dwells of one search enqueued back to back on one stream are searched in PAIRS (pcps_acquisition.cc:730-739 adds the second
 * |.|^2 to the first; here both are added in one pass and the grid is written once): the inverse passes of a dwell with
 * dwell counter < max_dwells are held back until the next call on the handle -- an accumulating dwell joins it, anything else
 * (fetch, grid read, new code, reset ...) first completes it alone or, for a reset, discards it.  Results and grids are bit-identical
 * to per-dwell processing; a gc_acq_flush() behind every dwell gives per-dwell processing (the evaluation after every dwell of pcps_acquisition.cc:747-755). */
gc_status gc_acq_dwell_enqueue(gc_acq* a, const void* dev_iq, void* stream);
gc_status gc_acq_fetch_results(gc_acq* a, gc_acq_result* host_results, void* stream);
/* Enqueues on `stream` whatever the enqueue-only calls have held back -- the inverse passes of a dwell waiting for a partner
 * and the statistics kernel of the last dwell (pcps_acquisition.cc:747-755 evaluates after every dwell) -- without copying
 * anything to the host and without synchronising: afterwards the device-side results are those of the last dwell, in stream
 * order.  gc_acq_fetch_results() = this + the copy + a stream synchronisation. */
gc_status gc_acq_flush(gc_acq* a, void* stream);
/* One dwell on the block of consumed_samples that starts at absolute sample first_index of the ring `s`
 * (the stream's format must equal gc_acq_set_input_format's; synchronous like gc_acq_dwell). */
gc_status gc_acq_dwell_stream(gc_acq* a, gc_stream* s, uint64_t first_index, gc_acq_result* host_results);
/* Copies satellite `sat`'s magnitude grid (num_doppler_bins * fft_size floats) to host. */
gc_status gc_acq_get_grid(gc_acq* a, int sat, float* host_grid);
/* Inspection of the engine's device-resident intermediates (tests, failure dumps; synchronises the context stream; natural element
 * order whatever the layout in HBM).  `what`:
 *   GC_ACQ_PEEK_WIPEOFF   index = Doppler bin: the wipe-off row exp(-j phase) of the ACTIVE grid, d_grid_doppler_wipeoffs[bin]
 *                         (pcps_acquisition.cc:296-310), fft_size complex values = 2 * fft_size floats
 *   GC_ACQ_PEEK_SPECTRUM  index = Doppler bin: FFT(x * wipeoff[bin]) of the last dwell's block (of the first block of a dwell
 *                         pair processed together) (:721), 2 * fft_size floats
 *   GC_ACQ_PEEK_CODE      index = satellite slot: conj(FFT(code)) (d_fft_codes, :272-273), 2 * fft_size floats
 *   GC_ACQ_PEEK_ROW_MAX   index = satellite slot: per Doppler bin the maximum of the grid row and its position as the statistics
 *                         kernel sees them (the column pass's block maxima, combined): 2 * num_doppler_bins floats (value, index) */
enum { GC_ACQ_PEEK_WIPEOFF = 0, GC_ACQ_PEEK_SPECTRUM = 1, GC_ACQ_PEEK_CODE = 2, GC_ACQ_PEEK_ROW_MAX = 3 };
gc_status gc_acq_peek(gc_acq* a, int what, int index, float* host_out);

#define GC_ABI_CHECK() \
    gc_abi_check(sizeof(gc_epoch_params), sizeof(gc_loop_conf), sizeof(gc_loop_record), sizeof(gc_loop_sync_conf), sizeof(gc_acq_conf), sizeof(gc_acq_result))

#ifdef __cplusplus
}
#endif
#endif /* GNSSCORR_H */
