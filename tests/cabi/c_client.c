/* c_client.c -- a plain C99 consumer of include/adsb_hip.h: proves the drop-in boundary needs nothing but the
 * header and libadsb_hip.so (no Python, no torch, no C++).  Test infrastructure.
 *
 *   c_client <fs> <threshold> <iq.f32> <expected.rec>   complex64 IQ file in, 32-byte adsb_burst records expected
 * exit 0 = identical, 2 = mismatch, 3 = no HIP device (adsb_create -> -ENODEV), 4 = usage / IO, 5 = ABI misuse
 * checks failed. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adsb_hip.h"

static void* slurp(const char* path, size_t* bytes) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(n > 0 ? (size_t)n : 1);
  if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
  fclose(f);
  *bytes = (size_t)(n > 0 ? n : 0);
  return p;
}

int main(int argc, char** argv) {
  if (sizeof(adsb_burst) != 32 || adsb_abi_version() != ADSB_ABI_VERSION) return 5;
  if (argc < 5) { fprintf(stderr, "usage: c_client fs threshold iq.f32 expected.rec\n"); return 4; }
  const double fs = atof(argv[1]);
  const float thr = (float)atof(argv[2]);
  adsb_ctx* c = NULL;
  /* argument checking happens before any device is touched */
  if (adsb_create(2.5e6, thr, 0, 0, &c) != -EINVAL || c != NULL) return 5;
  int rc = adsb_create(fs, thr, 0, 0, &c);
  if (rc == -ENODEV) { fprintf(stderr, "no HIP device: adsb_create -> -ENODEV (there is no CPU fallback)\n"); return 3; }
  if (rc != 0 || !c) { fprintf(stderr, "adsb_create: %d\n", rc); return 4; }

  size_t nb = 0, eb = 0;
  float* iq = (float*)slurp(argv[3], &nb);
  adsb_burst* want = (adsb_burst*)slurp(argv[4], &eb);
  if (!iq || !want) { fprintf(stderr, "cannot read inputs\n"); return 4; }
  const int64_t n = (int64_t)(nb / 8);
  const int32_t nwant = (int32_t)(eb / sizeof(adsb_burst));

  /* too small an output array: -ENOSPC and the required count */
  int32_t n_out = -1;
  adsb_burst one;
  rc = adsb_process_iq(c, iq, n, 0, &one, 1, &n_out);
  if (nwant > 1 && (rc != -ENOSPC || n_out != nwant)) { fprintf(stderr, "ENOSPC path: rc %d n_out %d\n", rc, n_out); return 5; }
  if (nwant > 1 && strlen(adsb_last_error(c)) == 0) return 5;

  adsb_burst* got = (adsb_burst*)calloc((size_t)nwant + 1, sizeof(adsb_burst));
  /* pinned input buffer: the DMA fast path */
  void* pinned = NULL;
  if (adsb_host_alloc(&pinned, nb ? nb : 1) != 0) return 4;
  memcpy(pinned, iq, nb);
  rc = adsb_process_iq(c, (const float*)pinned, n, 0, got, nwant + 1, &n_out);
  if (rc != 0) { fprintf(stderr, "adsb_process_iq: %d (%s)\n", rc, adsb_last_error(c)); return 4; }
  int bad = (n_out != nwant);
  for (int32_t i = 0; !bad && i < nwant; ++i) {
    /* compare what the reference defines (offset, SNR inputs, bits, PDU/no PDU); the oracle file carries no
     * parity pre-filter bits */
    bad = got[i].offset != want[i].offset || memcmp(&got[i].peak, &want[i].peak, 4) || memcmp(&got[i].median, &want[i].median, 4) ||
          memcmp(got[i].bits, want[i].bits, 14) || ((got[i].flags ^ want[i].flags) & ADSB_BURST_DEMOD);
    if (!bad && (got[i].flags & ADSB_BURST_DEMOD)) {
      int32_t df = -1, nbits = -1;
      const uint32_t syn = adsb_mode_s_syndrome(got[i].bits, &df, &nbits);
      const int pi = df == 11 || df == 17 || df == 18 || df == 19;
      if ((uint32_t)df != ADSB_BURST_DF(got[i].flags) || ((got[i].flags & ADSB_BURST_PARITY_OK) != 0) != (pi && syn == 0)) bad = 1;
    }
  }
  /* the same through the float |IQ|^2 entry (the framer's own input type) computed here in C */
  float* x = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  for (int64_t i = 0; i < n; ++i) {
    volatile float a = iq[2 * i] * iq[2 * i], b = iq[2 * i + 1] * iq[2 * i + 1];
    x[i] = a + b;
  }
  int32_t n2 = -1;
  rc = adsb_process_mag2(c, x, n, 0, NULL, 0, &n2);
  const adsb_burst* last = NULL;
  int32_t nlast = -1;
  if (rc != 0 || n2 != nwant || adsb_last_result(c, &last, &nlast) != 0 || nlast != nwant ||
      (nwant > 0 && memcmp(last, got, (size_t)nwant * sizeof(adsb_burst)) != 0)) bad = 1;

  /* host-fed pipelined submission (examples/adsb_rx.py:113-126,180-196 topology): three chunks in flight from the
   * page-locked buffer, collected in order, each equal to the blocking result shifted by its stream offset */
  {
    int32_t tk[3] = {-1, -1, -1};
    for (int k = 0; k < 3 && !bad; ++k)
      if (adsb_submit_format_host(c, ADSB_FMT_FC32, pinned, n, (int64_t)k * 1000, &tk[k]) != 0) bad = 1;
    int32_t t4 = -1;
    if (!bad && adsb_submit_format_host(c, ADSB_FMT_FC32, pinned, n, 0, &t4) != -EBUSY) bad = 1;   /* every slot in flight */
    adsb_burst* got2 = (adsb_burst*)calloc((size_t)nwant + 1, sizeof(adsb_burst));
    for (int k = 0; k < 3 && !bad; ++k) {
      int32_t m = -1;
      if (adsb_wait(c, tk[k], got2, nwant + 1, &m) != 0 || m != nwant) { bad = 1; break; }
      for (int32_t i = 0; i < nwant && !bad; ++i)
        bad = got2[i].offset != got[i].offset + (int64_t)k * 1000 || memcmp(got2[i].bits, got[i].bits, 14) ||
              got2[i].flags != got[i].flags || memcmp(&got2[i].median, &got[i].median, 4);
    }
    free(got2);
  }
  /* ABI 4: the stream as overlapped time shards, planned, pipelined and stitched inside the library (one C call), from a
   * DEVICE-resident copy of the stream (ABI 5: adsb_device_alloc / _upload -- this client does not link HIP): 1, 3 and 7
   * shards, each identical to the blocking result; the tiling itself (adsb_shard_bounds) covers the stream. */
  if (n >= 4096) {
    adsb_burst* got3 = (adsb_burst*)calloc((size_t)nwant + 1, sizeof(adsb_burst));
    void* d_iq = NULL;
    if (adsb_device_alloc(c, &d_iq, nb) != 0 || !d_iq || ((size_t)d_iq & 15u) || adsb_device_upload(c, d_iq, iq, nb) != 0) bad = 1;
    if (adsb_device_alloc(c, NULL, nb) != -EINVAL || adsb_device_upload(c, NULL, iq, nb) != -EINVAL) bad = 1;
    const int32_t shard_counts[3] = {1, 3, 7};
    for (int k = 0; k < 3 && !bad; ++k) {
      int32_t m = -1;
      const int32_t S = shard_counts[k];
      if (adsb_process_sharded_device(c, ADSB_FMT_FC32, d_iq, n, 0, S, got3, nwant + 1, &m) != 0 || m != nwant) { bad = 1; break; }
      for (int32_t i = 0; i < nwant && !bad; ++i)
        bad = got3[i].offset != got[i].offset || memcmp(got3[i].bits, got[i].bits, 14) || memcmp(&got3[i].peak, &got[i].peak, 8) ||
              ((got3[i].flags ^ got[i].flags) & (uint16_t)~ADSB_BURST_HEAD);
      int64_t prev_hi = 0;
      for (int32_t g = 0; g < S && !bad; ++g) {
        int64_t olo = -1, ohi = -1, lo = -1, hi = -1;
        if (adsb_shard_bounds(n, S, g, (int)(fs / 1e6), 4096, &olo, &ohi, &lo, &hi) != 0 || olo != prev_hi || lo > olo || hi < ohi || (lo & 3)) bad = 1;
        prev_hi = ohi;
      }
      if (!bad && prev_hi != n) bad = 1;
    }
    int32_t m0 = -1;
    if (!bad && nwant > 1 && (adsb_process_sharded_device(c, ADSB_FMT_FC32, d_iq, n, 0, 3, got3, 1, &m0) != -ENOSPC || m0 != nwant)) bad = 1;
    if (!bad && adsb_process_sharded_device(c, ADSB_FMT_FC32, d_iq, n, 0, 0, got3, nwant + 1, &m0) != -EINVAL) bad = 1;
    if (adsb_device_free(c, d_iq) != 0 || adsb_device_free(c, NULL) != 0) bad = 1;

    /* ABI 5: one process, N contexts (here: three on device 0 -- on an 8-GPU node one per device), ONE host buffer, one
     * stitched list: adsb_process_sharded_multi, from the page-locked copy and from the pageable one, 1 and 2 shards per
     * context; identical to the blocking result every time */
    {
      adsb_ctx* cs[3] = {c, NULL, NULL};
      if (adsb_create(fs, thr, 0, 0, &cs[1]) != 0 || adsb_create(fs, thr, 0, 0, &cs[2]) != 0) bad = 1;
      for (int nc = 1; nc <= 3 && !bad; nc += 2) {
        for (int spc = 1; spc <= 2 && !bad; ++spc) {
          const void* src = (spc == 1) ? (const void*)pinned : (const void*)iq;
          adsb_multi_stats ms;
          int32_t m = -1;
          memset(got3, 0, ((size_t)nwant + 1) * sizeof(adsb_burst));
          const int r = adsb_process_sharded_multi((adsb_ctx* const*)cs, nc, ADSB_FMT_FC32, src, n, 0, spc, got3, nwant + 1, &m, &ms);
          if (r != 0 || m != nwant || ms.contexts != nc || ms.shards != nc * spc || !(ms.wall_s > 0)) { fprintf(stderr, "multi: rc %d n %d (%s)\n", r, m, adsb_last_error(c)); bad = 1; break; }
          for (int32_t i = 0; i < nwant && !bad; ++i)
            bad = got3[i].offset != got[i].offset || memcmp(got3[i].bits, got[i].bits, 14) || memcmp(&got3[i].peak, &got[i].peak, 8) ||
                  ((got3[i].flags ^ got[i].flags) & (uint16_t)~ADSB_BURST_HEAD);
        }
      }
      if (!bad && nwant > 1 && (adsb_process_sharded_multi((adsb_ctx* const*)cs, 3, ADSB_FMT_FC32, pinned, n, 0, 1, got3, 1, &m0, NULL) != -ENOSPC || m0 != nwant)) bad = 1;
      adsb_ctx* const c2 = cs[2];
      cs[2] = cs[1];   /* a context listed twice */
      if (!bad && adsb_process_sharded_multi((adsb_ctx* const*)cs, 3, ADSB_FMT_FC32, pinned, n, 0, 1, got3, nwant + 1, &m0, NULL) != -EINVAL) bad = 1;
      cs[2] = NULL;
      if (!bad && adsb_process_sharded_multi((adsb_ctx* const*)cs, 3, ADSB_FMT_FC32, pinned, n, 0, 1, got3, nwant + 1, &m0, NULL) != -EINVAL) bad = 1;
      adsb_destroy(cs[1]);
      adsb_destroy(c2);
    }
    free(got3);
  }
  /* opt-in confidence ratios (demod.py:97-101): a context without the flag refuses, one with it returns n x 112 floats
   * whose sign test reproduces the hard bits (bit = bit1_amp > bit0_amp  <=>  ratio > 1 for positive amplitudes) */
  {
    const float* ratio = NULL;
    int32_t nr = -1;
    if (adsb_last_confidence(c, &ratio, &nr) != -EINVAL) bad = 1;
    adsb_ctx* cc = NULL;
    if (adsb_create(fs, thr, 0, ADSB_FLAG_CONFIDENCE, &cc) != 0) bad = 1;
    else {
      int32_t m = -1;
      if (adsb_process_iq(cc, (const float*)pinned, n, 0, NULL, 0, &m) != 0 || m != nwant ||
          adsb_last_confidence(cc, &ratio, &nr) != 0 || nr != nwant || (nwant > 0 && !ratio)) bad = 1;
      for (int32_t i = 0; i < nwant && !bad; ++i) {
        if (!(got[i].flags & ADSB_BURST_DEMOD)) continue;
        for (int k = 0; k < 112 && !bad; ++k) {
          const int bit = (got[i].bits[k >> 3] >> (7 - (k & 7))) & 1;
          const float q = ratio[(size_t)i * 112 + k];
          if (q == q && q != 1.0f && (q > 1.0f) != (bit != 0)) bad = 1;   /* NaN (0/0) and exact ties carry no sign */
        }
      }
      adsb_destroy(cc);
    }
  }

  printf("%d bursts, %s\n", (int)n_out, bad ? "MISMATCH" : "identical");
  adsb_host_free(pinned);
  adsb_destroy(c);
  free(got); free(x); free(iq); free(want);
  return bad ? 2 : 0;
}
