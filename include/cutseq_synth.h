/*
 * cutseq_synth.h -- C ABI of the seeded synthetic read generator, host and device form.
 *
 * Bench / test infrastructure, NOT part of the trimming boundary (that is cutseq_hip.h): the reference ships no
 * generator -- its only inputs are test/input_R{1,2}.fq.gz -- so nothing here replaces a reference interface.  The
 * generator is counter-based: pair i of a call is global pair `first_index + i`, its bytes depend on (seed, global
 * index) only, so any split of an index range over calls, threads, ranks or devices yields the same bytes, and
 *
 *   csh_synth_pairs   (libcutseq_host.so,  csrc/cutseq_host.c)     host arrays, n_threads pthreads
 *   csd_synth_pairs   (libcutseq_synth.so, csrc/synth_device.hip)  DEVICE arrays, one lane per pair, on `stream`
 *
 * write identical bytes (tests/test_gpu_synth.py).  Arrays: seq / qual [n][stride] bytes (bases / ASCII qualities,
 * zero-padded behind read_len; stride a multiple of 4), len [n] uint16.  Mate 2 arrays may be NULL iff single_end.
 */
#ifndef CUTSEQ_SYNTH_H
#define CUTSEQ_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct csh_synth_params {
  uint32_t read_len;
  uint32_t stride;
  uint64_t seed;
  uint64_t first_index;
  const char *p5_fw, *p7_fw, *p5_rc, *p7_rc; /* NUL-terminated, host memory */
  const char *inline5, *inline3;
  int32_t umi5, umi3, mask5, mask3;
  int32_t strand; /* +1, -1, 0 */
  int32_t single_end;
  double adapter_fraction; /* inserts shorter than the read: 3' adapter visible             */
  double partial_fraction; /* inserts that leave only a 3..19 nt adapter prefix in the read */
  double poly_fraction;    /* poly-A/T stretch of 10..40 nt at the insert end                */
  double art5_fraction;    /* 5' adapter artefact in front of the read                       */
  double sub_rate;         /* per-base substitution                                          */
  double indel_frac;       /* reads with an adapter that get one indel inside it             */
  double n_rate;           /* per-base N                                                     */
} csh_synth_params;

/* host form: 0, or -1 for parameters it rejects */
int csh_synth_pairs(const csh_synth_params *p, uint32_t n, uint8_t *seq1, uint8_t *qual1, uint16_t *len1, uint8_t *seq2,
                    uint8_t *qual2, uint16_t *len2, int n_threads);

/* device form: the arrays are device pointers of the current device, the kernel is enqueued on `stream` (a hipStream_t,
   NULL = the default stream) and the call returns without waiting.  0, -1 (parameters rejected) or -2 (launch failed);
   csd_last_error() names the reason (thread-local). */
int csd_synth_pairs(const csh_synth_params *p, uint64_t n, void *seq1, void *qual1, void *len1, void *seq2, void *qual2,
                    void *len2, void *stream);
const char *csd_last_error(void);
int csd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
