/* adsb_hip.h -- C ABI of libadsb_hip.so: the MI355X (gfx950) replacement for the framer + demod hot
 * path of mhostetter/gr-adsb.  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * The reference has no native code and therefore no FFI of its own: its two blocks are Python
 * gr.sync_block subclasses.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repo root):
 *
 *   adsb_create / adsb_destroy      framer.__init__ python/adsb/framer.py:37-65, demod.__init__ python/adsb/demod.py:35-54
 *   adsb_set_threshold              framer.set_threshold           python/adsb/framer.py:68-69
 *   adsb_framer_work[_passthrough]  framer.work()                  python/adsb/framer.py:72-182 (_passthrough: incl. :181 out0[:] = in0)
 *   adsb_demod_work                 demod.work()                   python/adsb/demod.py:57-136
 *   adsb_process_iq[_device]        complex_to_mag_squared -> framer -> demod as wired in
 *                                   examples/adsb_rx.py:180-196 (one canonical work() call per block)
 *   adsb_process_mag2[_device]      the same chain from the framer's float input onwards
 *   adsb_submit_*_device / adsb_wait   the same, up to ADSB_MAX_IN_FLIGHT calls in flight (no reference counterpart: pipelining)
 *   adsb_submit_format_host         the same fed from host memory: the SDR source -> framer chain of examples/adsb_rx.py:113-126,180-196
 *   adsb_last_confidence            demod.bit_confidence           python/adsb/demod.py:97-101
 *   adsb_device_alloc / _free / _upload   (no reference counterpart: device memory for C / ctypes clients of the *_device entries)
 *   adsb_shard_device / adsb_shard_fixup / adsb_stitch   (no reference counterpart: overlapped time shards, host stitch)
 *   adsb_process_sharded_multi      the single process of examples/adsb_rx.py:242-268 (one flowgraph, one IQ source) fed to
 *                                   N devices: one host ring in, one stitched burst list out (ABI 5)
 *
 * Conventions: the caller owns every buffer it passes; the library owns device memory, pinned staging
 * and one HIP stream per context.  A context is single-threaded; different contexts may be used
 * concurrently.  Return value 0 = success, negative errno otherwise; when an output array is too
 * small the call returns -ENOSPC and *n_out holds the required count.  A threshold change takes
 * effect at the next call.  There is no CPU fallback: without a usable HIP device adsb_create fails.
 */
#ifndef ADSB_HIP_H
#define ADSB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADSB_ABI_VERSION 5
#define ADSB_MAX_SPS 100 /* highest sample rate accepted: 100 Msps (tested up to and including it against the reference) */
#ifndef ADSB_MAX_IN_FLIGHT
#define ADSB_MAX_IN_FLIGHT 3 /* adsb_submit_* calls that may be pending at once */
#endif

/* Input sample formats (the `format` / `fmt` argument).  The reference's flowgraph feeds complex64 from the
 * SDR source through complex_to_mag_squared (examples/adsb_rx.py:116,180); the integer formats are the same
 * samples as the SDR hardware delivers them, converted exactly on the device. */
#define ADSB_FMT_FC32 0 /* interleaved float32 I,Q           8 B/sample */
#define ADSB_FMT_MAG2 1 /* float32 |IQ|^2 (the framer's own input type, framer.py:38)  4 B/sample */
#define ADSB_FMT_SC16 2 /* interleaved int16 I,Q             4 B/sample   component = f32(i16) * scale   (default 1/32768) */
#define ADSB_FMT_SC8 3  /* interleaved int8 I,Q              2 B/sample   component = f32(i8) * scale    (default 1/128) */
#define ADSB_FMT_CU8 4  /* interleaved uint8 I,Q, offset binary (RTL-SDR)  2 B/sample
                         * component = f32(2*u8 - 255) * scale, i.e. (u8 - 127.5) * 2*scale, exact (default 1/255).
                         * A power-of-two scale -- e.g. 2^-8: the (u8 - 127.5) / 128 of the usual RTL-SDR front ends -- runs, like a
                         * power-of-two int8 scale, an instance of the streaming kernel that squares with integer dot products:
                         * same bits, 4-6 % faster */
#define ADSB_FMT_COUNT 5

/* adsb_create flags */
#define ADSB_FLAG_TIMING 1u /* bracket the detect kernel with HIP events (adsb_get_stats) */
/* Opt-in extension, NOT the reference's behaviour: the re-trigger gate holds for the length of the
 * burst it accepted -- 119*sps when the burst's first data bit is set (DF >= 16: a 112-bit reply), 63*sps otherwise --
 * instead of always assuming a short reply (framer.py:163-165), so the second half of a long reply can no longer
 * raise false tags.  Applies to adsb_process_* / adsb_submit_* / adsb_shard_*; adsb_framer_work (the exact GNU Radio
 * emulation) ignores it.  Records of such a context carry ADSB_BURST_LONG_HINT, which adsb_shard_fixup / adsb_stitch
 * honour. */
#define ADSB_FLAG_LONG_AWARE_GATE 2u
/* Opt-in: keep what demod.work() keeps as self.bit_confidence (demod.py:97-101, never published by the reference) for
 * every delivered burst of the whole-buffer entry points -- the float32 ratios bit1_amp / bit0_amp, see
 * adsb_last_confidence.  Costs one extra small kernel and copy per call. */
#define ADSB_FLAG_CONFIDENCE 4u
/* Run the sparse tail of a pass on the compute stream behind its k_detect instead of on a second stream beside the
 * next pass's k_detect (profiling aid: serial kernels; about 15 % less throughput with several calls in flight). */
#define ADSB_FLAG_SINGLE_STREAM 8u
/* Pipelined use (adsb_submit_*): let the sparse tail of a pass (ordering, gate, compaction of the burst records) run on
 * the GPU BESIDE the next pass's streaming kernel instead of after it: results arrive about one pass earlier and two
 * calls in flight suffice, for 1-2 % less throughput.  Default off = maximum throughput. */
#define ADSB_FLAG_LOW_LATENCY 16u
/* adsb_framer_work also slices the 112 bits of every tag whose burst ends inside the call's input
 * (offset + 119*sps + sps/2 < nitems_written + N, the rule of demod.py:82 applied to the framer's own chunk): such tags
 * come back with ADSB_BURST_DEMOD, bits and the parity pre-filter flags -- the one device pass a framer/demod pair of
 * one flowgraph needs (gr_adsb_amd.blocks.demod(fs, framer=...)); default off: tags carry offset / peak / median only. */
#define ADSB_FLAG_FRAMER_SLICES 32u
/* Host-side NUMA placement is ON by default: the context looks up the NUMA node and the cpus local to its GPU's PCI device
 * (/sys/bus/pci/devices/<bdf>/numa_node, local_cpulist), allocates its page-locked buffers (staging ring, result and
 * summary buffers) on that node and runs its copy threads on those cpus -- on a two-socket 8-GPU node (one process per GPU,
 * SURVEY.md §8e) half of the host-fed traffic would otherwise cross the socket interconnect.  This flag turns it off
 * (adsb_numa_info still reports what was found).  No reference counterpart: the reference is one Python thread. */
#define ADSB_FLAG_NO_NUMA_BINDING 64u

/* adsb_burst.flags */
#define ADSB_BURST_DEMOD 1u /* eob inside the demod input: bits[] valid, a PDU is published (demod.py:82) */
#define ADSB_BURST_KEPT 2u  /* passed the framer's re-trigger gate (framer.py:121) */
#define ADSB_BURST_HEAD 16u /* shard mode: part of the shard's head region (see adsb_shard_device) */
/* Mode S parity pre-filter, computed on the device for every burst with ADSB_BURST_DEMOD:
 * what the decoder's first two steps (decoder.py:550-556 decode_header, :560-688 check_parity) will find,
 * so a consumer can drop garbage PDUs before they reach the (scalar, per-message) decoder.  Advisory: the
 * bits and every other field are unchanged, and the reference-compatible blocks publish all PDUs. */
#define ADSB_BURST_PARITY_OK 32u /* DF 11/17/18/19 and crc(bits[0:L-24]) == bits[L-24:L] (decoder.py:625,679) */
#define ADSB_BURST_LONG 64u      /* DF 16-21/24: 112-bit reply (decoder.py:636,669); else the 56-bit reading */
#define ADSB_BURST_KNOWN_DF 128u /* DF is one check_parity() handles: 0,4,5,11,16,17,18,19,20,21,24 */
#define ADSB_BURST_DF_SHIFT 8    /* (flags >> 8) & 31 = downlink format (decoder.py:551) */
#define ADSB_BURST_DF(flags) (((flags) >> ADSB_BURST_DF_SHIFT) & 31u)
#define ADSB_BURST_LONG_HINT 0x2000u /* long-aware contexts only: this burst holds the gate for 119*sps */

typedef struct adsb_ctx adsb_ctx;

/* One detected burst == one "burst" stream tag (framer.py:168-174) plus, when ADSB_BURST_DEMOD is set,
 * the payload of the PDU demod would publish for it (demod.py:104-110).  32 bytes, little endian. */
typedef struct adsb_burst {
  int64_t offset;   /* absolute stream offset of the tag: centre of the first preamble pulse */
  float peak;       /* in0[pulse_idx]                       (framer.py:157) */
  float median;     /* np.median of the <=100 samples before (framer.py:157-159) */
  uint8_t bits[14]; /* 112 hard bits, first bit = MSB of bits[0] (demod.py:94-95) */
  uint16_t flags;
} adsb_burst;

typedef struct adsb_stats {
  uint64_t detect_launches;  /* k_detect launches timed */
  double detect_ms;          /* sum of their HIP-event durations */
  uint64_t detect_samples;   /* samples those launches covered */
  uint64_t detect_bytes;     /* algorithmic bytes: samples x the format's bytes per sample */
  uint64_t calls;
  uint64_t retries;          /* record-capacity regrowths */
  uint64_t longrun_calls;    /* calls that needed the long-pulse kernel */
  uint64_t detect_grid;      /* workgroups of the last k_detect launch */
  uint64_t blocks_per_cu;    /* resident k_detect workgroups per CU (occupancy query): four wavefronts each, one for the 8-bit formats */
  double detect_gap_ms;      /* sum of idle gaps on the compute stream between consecutive timed k_detect launches */
  uint64_t detect_gaps;      /* number of gaps summed */
  uint64_t longrun_pulses;   /* pulses longer than k_detect's LDS window, handled by the long-pulse kernel (sum over calls) */
  uint64_t poll_fallbacks;   /* ABI 3: small passes whose pass number did not appear within the short spin and were waited for
                              * by blocking on the stream instead (adsb_hip.hip: finish) */
  uint64_t shard_fallbacks;  /* ABI 4: shards of adsb_process_sharded_device whose head region ended inside a chain of overlapping
                              * bursts and were run a second time (larger head, then ungated + greedy gate) */
} adsb_stats;

int adsb_abi_version(void);

/* fs must be an even multiple of 1e6 (the reference asserts fs % 1e6 == 0, framer.py:44, and only
 * works for even sps) between 2e6 and ADSB_MAX_SPS * 1e6: otherwise -EINVAL.  2 / 4 / 8 / 20 Msps run
 * kernels with the preamble tap stride (sps / 2, framer.py:137) compiled in, every other rate ("2 Msps, 4 Msps, 6 Msps,
 * etc", README.md:17) the run-time-stride instances; all are pinned by reference vectors (tests/golden/R*.npz: 6, 10, 12,
 * 16, 24, 40 and 100 Msps).  The reference itself has no upper limit; above 100 Msps nothing is tested, so nothing is
 * accepted.  device = HIP ordinal. */
int adsb_create(double fs, float threshold, int device, uint32_t flags, adsb_ctx** out);
void adsb_destroy(adsb_ctx* ctx);
int adsb_set_threshold(adsb_ctx* ctx, float threshold);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own compute stream: device
 * input produced by work queued on that stream needs no host synchronisation before adsb_process_*_device /
 * adsb_submit_*.  With the context's own stream (the default) the caller synchronises producers of a device
 * buffer first.  -EBUSY while submitted calls are pending.  A context is used from one thread at a time;
 * different contexts are independent. */
int adsb_set_stream(adsb_ctx* ctx, void* hip_stream);
/* Host threads (1..64, the caller included) that copy PAGEABLE sources of adsb_submit_format_host into the pinned staging
 * ring; default 6 on hosts with >= 16 cpus.  Before the first pageable submission only (-EBUSY afterwards).  Page-locked
 * sources (adsb_host_alloc, adsb_host_register) never touch these threads: they are DMA'd where they lie. */
int adsb_set_copy_threads(adsb_ctx* ctx, int32_t threads);
/* memcpy split over the context's copy threads (the GNU Radio passthrough `out0[:] = in0` of multi-megabyte chunks:
 * framer.py:181, demod.py:135).  Plain host memory on both sides; blocking. */
int adsb_host_copy(adsb_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Order the NEXT call on this context -- blocking, adsb_submit_* or the sharded driver's first pass -- AFTER a HIP event of the
 * caller (hipEvent_t recorded on the stream that produces a device-resident input, e.g. a framework's current stream): a
 * device-side dependency, the host does not wait.  ABI 4: submitted passes run on one stream per pipeline slot, so the wait is
 * queued with that next call, on the stream its first operation runs on (a host-fed submission: the upload stream); a later submission that depends on the same producer asks again.  Up to four events may be pending.  The
 * event must stay alive until that next call has returned. */
int adsb_wait_for_event(adsb_ctx* ctx, void* hip_event);
/* Pending events are consumed by the next call that queues GPU work.  A call that returns before it queues anything (an
 * argument error, -EBUSY, n == 0, adsb_demod_work without tags) leaves them pending for the call after it -- the events must
 * stay alive until then, or be dropped with adsb_clear_pending_events (ABI 5; adsb_reset drops them too). */
int adsb_clear_pending_events(adsb_ctx* ctx);
/* Forget the framer's cross-call state (prev_in0 = 0, prev_eob = -1; framer.py:54,57) and any pending adsb_wait_for_event. */
int adsb_reset(adsb_ctx* ctx);
/* The framer's two words of cross-call state as the reference keeps them on the block (framer.py:54 `prev_in0`, :57
 * `prev_eob_idx`, both public attributes there): what adsb_framer_work carries between calls.  Either pointer may be NULL. */
int adsb_framer_state(adsb_ctx* ctx, float* prev_in0, int64_t* prev_eob_idx);

/* Canonical whole-buffer mode: ONE framer.work() call over n samples of a fresh stream (history =
 * 8*sps-1 zeros) followed by ONE demod.work() call over the same n samples with all tags delivered.
 * iq: n interleaved complex64 (2n floats).  abs_offset: stream offset of sample 0.  Output: the
 * kept bursts in stream order; bursts whose eob falls outside the buffer have ADSB_BURST_DEMOD clear
 * (tag emitted, PDU dropped: demod.py:130-133).  Stateless across calls. */
int adsb_process_iq(adsb_ctx* ctx, const float* iq_host, int64_t n, int64_t abs_offset,
                    adsb_burst* out, int32_t cap, int32_t* n_out);
int adsb_process_mag2(adsb_ctx* ctx, const float* mag2_host, int64_t n, int64_t abs_offset,
                      adsb_burst* out, int32_t cap, int32_t* n_out);
/* Page-locked host memory for IQ buffers handed to adsb_process_iq / adsb_process_mag2 / adsb_framer_work:
 * buffers allocated here (or any other pinned host memory) are DMA'd straight to the device; pageable
 * buffers are first copied into the context's own pinned staging buffer (about 3x slower end to end). */
int adsb_host_alloc(void** p, size_t bytes);
int adsb_host_free(void* p);
/* The same on the NUMA node of the context's GPU (see ADSB_FLAG_NO_NUMA_BINDING): what a feeder should fill its IQ ring
 * from, so that the H2D DMA reads local memory.  Freed with adsb_host_free. */
int adsb_host_alloc_near(adsb_ctx* ctx, void** p, size_t bytes);
/* Where the context's host side lives: NUMA node of the GPU's PCI device (-1: unknown / not bound), the cpus local to it
 * as sysfs prints them ("0-63,128-191", "" if unknown) and the device's PCI address ("0000:c1:00.0").  Any pointer may be
 * null.  For reports (bench.py prints it per rank) and for callers that want to pin their own feeder threads. */
int adsb_numa_info(adsb_ctx* ctx, int32_t* node, char* cpulist, size_t cpulist_cap, char* pci_bdf, size_t bdf_cap);
/* Page-lock a buffer the caller already owns (e.g. the ring an SDR driver or a file mapping fills), so that
 * adsb_process_* / adsb_submit_format_host DMA it where it lies instead of copying it through the context's staging
 * chunks (about half the rate).  Registration costs milliseconds: do it once per buffer, not per call; unregister before
 * freeing the memory.  Thin wrappers over hipHostRegister / hipHostUnregister for callers that do not link HIP. */
int adsb_host_register(void* p, size_t bytes);
int adsb_host_unregister(void* p);

/* Same, input already in HBM (16-byte aligned device pointer).  out may be NULL: the result stays in
 * the context's pinned buffer, see adsb_last_result. */
int adsb_process_iq_device(adsb_ctx* ctx, const void* d_iq, int64_t n, int64_t abs_offset,
                           adsb_burst* out, int32_t cap, int32_t* n_out);
int adsb_process_mag2_device(adsb_ctx* ctx, const void* d_mag2, int64_t n, int64_t abs_offset,
                             adsb_burst* out, int32_t cap, int32_t* n_out);
/* int16 IQ (interleaved I,Q shorts, 4 B/sample: the SDR's native wire format).  Each
 * component becomes float32 exactly and is multiplied by `scale` (float32, one rounded multiply; default
 * 1/32768) before |IQ|^2; everything downstream is identical to the complex64 path. */
int adsb_set_iq16_scale(adsb_ctx* ctx, float scale);
int adsb_process_iq16(adsb_ctx* ctx, const int16_t* iq16_host, int64_t n, int64_t abs_offset,
                      adsb_burst* out, int32_t cap, int32_t* n_out);
int adsb_process_iq16_device(adsb_ctx* ctx, const void* d_iq16, int64_t n, int64_t abs_offset,
                             adsb_burst* out, int32_t cap, int32_t* n_out);
/* Any format by number (ADSB_FMT_*): the entry points above are adsb_process_format[_device] with format 0, 1, 2.
 * Integer components become float32 exactly and are multiplied by the format's scale (one rounded float32
 * multiply) before |IQ|^2 = re*re + im*im with separately rounded products; downstream is identical.
 * adsb_set_format_scale: format must be one of the integer formats (-EINVAL otherwise). */
int adsb_set_format_scale(adsb_ctx* ctx, int format, float scale);
int adsb_process_format(adsb_ctx* ctx, int format, const void* host, int64_t n, int64_t abs_offset,
                        adsb_burst* out, int32_t cap, int32_t* n_out);
int adsb_process_format_device(adsb_ctx* ctx, int format, const void* d_data, int64_t n, int64_t abs_offset,
                               adsb_burst* out, int32_t cap, int32_t* n_out);
int adsb_last_result(adsb_ctx* ctx, const adsb_burst** bursts, int32_t* n);
/* ADSB_FLAG_CONFIDENCE contexts: *ratio -> n x 112 float32 in the context's pinned memory, row t = bit1_amp / bit0_amp
 * of burst t of the last finished call (demod.py:91-101: 10*log10 of it is bit_confidence; +-inf / NaN where the
 * reference has them); rows of bursts without ADSB_BURST_DEMOD are zero.  Valid until the same pipeline slot is
 * used again.  -EINVAL on a context created without the flag. */
int adsb_last_confidence(adsb_ctx* ctx, const float** ratio, int32_t* n);

/* Asynchronous form of adsb_process_*_device: submit queues the whole device pipeline on the context's
 * streams and returns a ticket (0 .. ADSB_MAX_IN_FLIGHT-1) at once; adsb_wait blocks for that call, copies
 * its bursts to pinned host memory on a copy stream and delivers them like adsb_process_*.  With later
 * calls submitted before waiting for call i, the streaming kernel of call i+1 runs back to back with that
 * of call i while call i's tail kernels, its PCIe copy and all host work proceed beside it.  The input
 * buffer must stay valid and unchanged until adsb_wait returns.  At most ADSB_MAX_IN_FLIGHT calls pending
 * (-EBUSY otherwise); results must be collected in submission order. */
int adsb_submit_iq_device(adsb_ctx* ctx, const void* d_iq, int64_t n, int64_t abs_offset, int32_t* ticket);
int adsb_submit_mag2_device(adsb_ctx* ctx, const void* d_mag2, int64_t n, int64_t abs_offset, int32_t* ticket);
int adsb_submit_iq16_device(adsb_ctx* ctx, const void* d_iq16, int64_t n, int64_t abs_offset, int32_t* ticket);
int adsb_submit_format_device(adsb_ctx* ctx, int format, const void* d_data, int64_t n, int64_t abs_offset, int32_t* ticket);
/* Host-fed streaming, the topology of examples/adsb_rx.py:113-126,180-196 (SDR source -> ... -> framer -> demod) with
 * the chunks in host memory: the samples are uploaded on a dedicated stream into the ticket's own device buffer, so
 * with several calls in flight the upload of chunk i+1 runs beside the kernels of chunk i and the record download of
 * chunk i-1 (PCIe both ways, compute in between).  A page-locked source (adsb_host_alloc) is DMA'd where it lies and
 * must stay valid until adsb_wait returns; a pageable source is copied through pinned chunks before the call returns
 * (the call then takes as long as that copy).  Results exactly as adsb_process_format. */
int adsb_submit_format_host(adsb_ctx* ctx, int format, const void* host, int64_t n, int64_t abs_offset, int32_t* ticket);
int adsb_submit_shard_device(adsb_ctx* ctx, int fmt, const void* d_data, int64_t n, int64_t origin, int64_t own_lo,
                             int64_t own_hi, int64_t stream_len, int32_t head_cands, int32_t* ticket);
int adsb_wait(adsb_ctx* ctx, int32_t ticket, adsb_burst* out, int32_t cap, int32_t* n_out);

/* GNU Radio sync-block emulation, framer.work(): in0 holds N + 8*sps - 1 floats of |IQ|^2 (history
 * first), exactly what the scheduler hands the Python block; nitems_written = nitems_written(0).
 * Emits the tags of this call (offset/peak/median; flags = KEPT) and carries prev_in0 / prev_eob_idx
 * inside ctx exactly like the reference (including its stale-state behaviour, framer.py:177-179). */
int adsb_framer_work(adsb_ctx* ctx, const float* in0, int64_t n_in0, int64_t N, int64_t nitems_written,
                     adsb_burst* tags, int32_t cap, int32_t* n_out);
/* The same plus the block's pass-through (framer.py:181 `out0[:] = in0[history:]`): out0 (N floats, may be NULL) is filled
 * on the host WHILE the device pass runs -- the copy of a multi-megabyte chunk no longer stands behind the pass (ABI 5). */
int adsb_framer_work_passthrough(adsb_ctx* ctx, const float* in0, int64_t n_in0, int64_t N, int64_t nitems_written, float* out0,
                                 adsb_burst* tags, int32_t cap, int32_t* n_out);

/* demod.work(): in0 = this call's n input floats, nitems_read = nitems_read(0) (== nitems_written(0)
 * for a sync block); tag_offsets = absolute offsets of the "burst" tags inside [nitems_read,
 * nitems_read+n).  bits112: ntags*112 bytes of 0/1 (the u8vector the PDU carries); ok[t] != 0 when the
 * burst was demodulated, 0 when it straddles the end of the chunk and is dropped (demod.py:82,130-133);
 * a non-zero ok[t] is ADSB_BURST_DEMOD | the pre-filter bits ADSB_BURST_PARITY_OK / _LONG / _KNOWN_DF.
 * ratio (optional, may be NULL): ntags*112 floats bit1_amp/bit0_amp; 10*log10 of it is
 * demod.bit_confidence (demod.py:101). */
int adsb_demod_work(adsb_ctx* ctx, const float* in0, int64_t n, int64_t nitems_read,
                    const int64_t* tag_offsets, int32_t ntags, uint8_t* bits112, uint8_t* ok, float* ratio);

/* Overlapped time shards (multi-GPU): the device buffer holds stream samples [origin, origin+n) of
 * which this shard owns the pulse rises in [own_lo, own_hi) (stream offsets).  stream_len = length of
 * the whole stream (for the end-of-stream rules); fmt = ADSB_FMT_*.
 *   head_cands == 0: returns EVERY matched preamble centre of the owned range, not gated (KEPT never
 *     set); adsb_stitch applies the gate over the concatenation of all shards.
 *   head_cands  > 0: the gate runs on the device as if the shard started a fresh stream (KEPT set), and
 *     the first head_cands centres of the shard are returned whether gated or not (HEAD set): with the
 *     previous shard's end-of-burst state adsb_shard_fixup then makes the result exact on the host, so
 *     ranks exchange 8 bytes instead of candidate lists.
 * -EOVERFLOW when a pulse or burst runs past the shard's halo. */
int adsb_shard_device(adsb_ctx* ctx, int fmt, const void* d_data, int64_t n, int64_t origin,
                      int64_t own_lo, int64_t own_hi, int64_t stream_len, int32_t head_cands,
                      adsb_burst* out, int32_t cap, int32_t* n_out);
/* The same from a host buffer (uploaded like adsb_process_*): for callers that receive the stream block by block
 * in host memory -- the chunk-invariant ("improved") GNU Radio blocks and file replay without torch.  origin may
 * be negative when the buffer starts with the zero history in front of a fresh stream.
 * shard_flags: ADSB_SHARD_DROP_OVERLONG = a pulse still high at the end of the buffer is left out of the result
 * (as framer.py:102-108 leaves out a pulse still high at the end of a call) instead of failing with -EOVERFLOW. */
#define ADSB_SHARD_DROP_OVERLONG 1u
int adsb_shard_host(adsb_ctx* ctx, int fmt, const void* host, int64_t n, int64_t origin, int64_t own_lo,
                    int64_t own_hi, int64_t stream_len, int32_t head_cands, uint32_t shard_flags,
                    adsb_burst* out, int32_t cap, int32_t* n_out);
/* The tiling every sharded caller uses (gr_adsb_amd/frontend.py: shard_plan; bench.py's ranks; file replay; the driver
 * below): shard g of n_shards over a stream of stream_len samples owns the pulse rises of [*own_lo, *own_hi) -- equal
 * ranges of a multiple of `align` samples -- and needs the samples [*lo, *hi): 100 + 8*sps + 4 of back halo (noise window,
 * framer.py:31,156; preamble span), *lo on a 16-byte boundary of every format, and 256 + 121*sps of forward halo (the longest
 * pulse followed, preamble + 112 bits: framer.py:165, demod.py:76).  Pure host arithmetic; 0 or -EINVAL. */
int32_t adsb_shard_bounds(int64_t stream_len, int32_t n_shards, int32_t g, int sps, int64_t align, int64_t* own_lo,
                          int64_t* own_hi, int64_t* lo, int64_t* hi);
/* ONE resident stream processed as `shards` overlapped time shards on THIS device -- BASELINE config 4's decomposition
 * (one 20 Msps stream tiled as N overlapped shards) run where there is one GPU, or many receivers' worth of mid-size
 * buffers multiplexed on it: the shards are planned (adsb_shard_bounds, align 4096), kept ADSB_MAX_IN_FLIGHT deep in the
 * pipeline, each detected and gated on the device as a fresh stream; the head of every shard is re-gated on the host with
 * the end-of-burst state carried from the shard in front of it (adsb_shard_fixup; a head that ends inside a chain of
 * overlapping bursts: the shard once more with the largest head, then ungated with the plain greedy gate).  The whole loop
 * is host C: no interpreter between two passes.  Result: bit-identical to adsb_process_format_device over the whole
 * buffer (records, order, flags except ADSB_BURST_HEAD, which is cleared).  out must hold the result (-ENOSPC with *n_out =
 * the number needed otherwise).  Replaces: one framer.work() + demod.work() over the stream (framer.py:72-182,
 * demod.py:57-136), like adsb_process_format_device; the tiling itself has no reference counterpart. */
int adsb_process_sharded_device(adsb_ctx* ctx, int format, const void* d_data, int64_t n, int64_t abs_offset,
                                int32_t shards, adsb_burst* out, int32_t cap, int32_t* n_out);
/* (d_data: memory the DEVICE can read -- device memory, e.g. adsb_device_alloc below, or page-locked host memory, which the
 * kernels then read over PCIe.  Every shard pass waits for the events of adsb_wait_for_event.  -EINVAL on an
 * ADSB_FLAG_CONFIDENCE context: the rows of adsb_last_confidence belong to the records of one pass.) */

/* ONE process, N devices, ONE host ring (ABI 5): the reference is a single process with a single IQ source
 * (examples/adsb_rx.py:242-268); this is that process with N GPUs behind it.  `host` holds n samples of `format` (page-locked
 * -- adsb_host_alloc[_near], adsb_host_register: DMA'd where they lie -- or pageable: through each context's staging ring);
 * ctxs[0..n_ctx) are contexts of the SAME rate, threshold, gate flag and format scale, normally one per device (several on
 * one device are allowed: that is how a one-GPU box tests it).  The stream is tiled into n_ctx * shards_per_ctx overlapped
 * time shards (adsb_shard_bounds, align 4096); context k takes shards [k*shards_per_ctx, (k+1)*shards_per_ctx).  Inside the
 * call one feeder thread per context -- on the cpus local to its GPU, within the process's own mask -- uploads shard i+1
 * beside the shard pass of i and the record download of i-1 (ADSB_MAX_IN_FLIGHT deep); the calling thread takes finished
 * shards in stream order, re-gates every head with the end-of-burst state carried over the seam (adsb_shard_fixup) and
 * appends the kept records to `out`; a head that ends inside a chain of overlapping bursts has its shard run again on its
 * own context (head 4096, then ungated + the plain greedy gate: exact in every case).  No collective, no second process:
 * one int64 crosses each seam, on the host.  Result: bit-identical to adsb_process_format over the whole buffer (records,
 * order, flags except ADSB_BURST_HEAD, cleared).  -ENOSPC with *n_out = the number needed when `out` is too small; on any
 * other error nothing stays in flight on any context and adsb_last_error(ctxs[0]) names the cause.  stats may be NULL. */
#define ADSB_MULTI_MAX_CTX 64
typedef struct adsb_multi_stats {
  int32_t contexts, shards, fallbacks, pad_;
  double wall_s;                         /* the whole call */
  double feeder_s[ADSB_MULTI_MAX_CTX];   /* context k's feeder thread: first upload queued -> last shard collected */
  int32_t device[ADSB_MULTI_MAX_CTX];    /* HIP ordinal of context k */
  int32_t numa_node[ADSB_MULTI_MAX_CTX]; /* NUMA node its pinned buffers and feeder live on (-1: unknown / not bound) */
} adsb_multi_stats;
int adsb_process_sharded_multi(adsb_ctx* const* ctxs, int32_t n_ctx, int format, const void* host, int64_t n,
                               int64_t abs_offset, int32_t shards_per_ctx, adsb_burst* out, int32_t cap, int32_t* n_out,
                               adsb_multi_stats* stats);
/* Device memory on the context's device for callers that do not link HIP (a C or ctypes client of the *_device entry
 * points): hipMalloc / hipFree / a blocking hipMemcpy host -> device.  16-byte alignment is guaranteed.  No reference
 * counterpart (the reference never leaves host memory). */
int adsb_device_alloc(adsb_ctx* ctx, void** d, size_t bytes);
int adsb_device_free(adsb_ctx* ctx, void* d);
int adsb_device_upload(adsb_ctx* ctx, void* d, const void* host, size_t bytes);
/* eob_in = (offset of the last burst kept before this shard) + 63*sps, or a very negative number for the
 * first shard.  Compacts recs in place to the exact kept list; -EAGAIN if the head region was too short
 * (call adsb_shard_device again with a larger head_cands, or with 0 and adsb_stitch). */
int adsb_shard_fixup(adsb_burst* recs, int32_t n, int sps, int64_t eob_in, int32_t* n_kept);
/* Host stitch: cands = shard outputs concatenated in stream order; applies the re-trigger gate
 * (framer.py:121-123,165) in place (sets KEPT) and compacts the kept bursts to the front. */
int adsb_stitch(adsb_burst* cands, int32_t n, int sps, int32_t* n_kept);

/* 10*log10(peak/median) + 1.6 in float32 (framer.py:157) with libm's log10f.  NumPy's float32 log10 is
 * a SIMD routine on some hosts, so the reference's SNR bits are host dependent; the Python shim finalises
 * SNR with NumPy from (peak, median) -- those two ARE bit exact -- and this helper is for C callers. */
float adsb_snr_db(float peak, float median);

/* Host helper for the address/parity formats (DF 0/4/5/16/20/21/24), whose check needs the consumer's
 * aircraft table: returns crc(bits[0:L-24]) ^ bits[L-24:L] -- the announced address `aa` of
 * decoder.py:577,647 (0 for a clean DF 11/17/18/19) -- for the DF's length L; *df = downlink format,
 * *nbits = 56 / 112, or 0 for a DF check_parity() does not know (the 56-bit reading is returned).
 * Pure host arithmetic on one 14-byte payload; out pointers may be NULL. */
uint32_t adsb_mode_s_syndrome(const uint8_t bits[14], int32_t* df, int32_t* nbits);

/* How one call over n_samples is cut on a device that keeps `resident_wavefronts` wavefronts of the streaming kernel
 * resident (adsb_stats.detect_grid / blocks_per_cu tell what a context uses: CUs x blocks_per_cu x 4, or x 1 for the 8-bit formats): *units chunks of
 * *samples_per_chunk samples each (the last may be shorter), one wavefront and one output list per chunk.  One resident
 * round is the floor; a bulk call runs up to eight rounds of shorter chunks (never shorter than 4096 samples) so that the
 * dispatcher evens out wavefronts that finish apart.  Pure host arithmetic (no device needed), the reference has no
 * counterpart: framer.py:83-174 walks the whole in0 in one Python loop.  Returns 0 or -EINVAL. */
int32_t adsb_plan_chunks(int64_t n_samples, int64_t resident_wavefronts, int64_t* units, int64_t* samples_per_chunk);

int adsb_get_stats(adsb_ctx* ctx, adsb_stats* out);
int adsb_reset_stats(adsb_ctx* ctx);
/* The duration (ms, HIP events on the compute stream) of every k_detect launch the context has timed since the last
 * adsb_reset_stats, oldest first -- the last 4096 of them; contexts created with ADSB_FLAG_TIMING only.  adsb_stats holds
 * their sum; this is the per-launch sequence (measurement aid: tools/launch_hist.py holds it against a rocprofv3 kernel
 * trace launch by launch).  No reference counterpart.  *n = durations written (<= cap). */
int adsb_detect_history(adsb_ctx* ctx, float* ms, int32_t cap, int32_t* n);
/* Text of the last error on this context ("" if none). */
const char* adsb_last_error(adsb_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ADSB_HIP_H */
