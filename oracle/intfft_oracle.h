/*
 * intfft_oracle.h -- CPU restatement of the intfftk fixed-point radix-2 FFT/IFFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under intfftk_amd/ (the product) may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference (hukenovs/intfftk) ships no golden vectors, no self-checking
 * testbench and no runnable model of the fixed-point arithmetic (the Octave model is double
 * precision; the RTL needs a VHDL simulator + Xilinx unisim, neither present).  This oracle
 * is therefore a restatement of the RTL text, cross-checked three ways (see oracle/README.md):
 * stream form == in-place form, C == independent Python twin, and numeric closeness to
 * numpy.fft.  Every function cites the reference file:line it follows.
 */
#ifndef INTFFT_ORACLE_H
#define INTFFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int64_t re, im; } orc_cplx;

/* Mirrors the VHDL generics of int_fftNk / int_ifftNk (src/vhdl/fft/int_fftNk.vhd:73-84). */
typedef struct {
    int log2n;      /* NFFT generic = log2 of the length                          */
    int data_width; /* DATA_WIDTH (input width of the first core)                  */
    int twdl_width; /* TWDL_WIDTH                                                  */
    int format;     /* 1 unscaled, 0 scaled                                        */
    int rndmode;    /* 0 truncate, 1 round-half-up (scaled only)                   */
    int xser;       /* 0 "OLD" (DSP48E1), 1 "NEW" (DSP48E2)                        */
    int use_fly;    /* 1 butterflies on, 0 bypass (permutation network only)       */
} orc_params;

enum { ORC_FWD = 0, ORC_INV = 1, ORC_PAIR = 2 };
/* memory index m of a frame -> logical sample index (see include/intfft.h) */
enum { ORC_NATURAL = 0, ORC_BITREV = 1, ORC_HALVES = 2, ORC_BITREV_LANES = 3 };
/* complex-multiplier regimes, int_cmult_dsp48.vhd:182-434 */
enum { ORC_SNGL = 0, ORC_DBL18, ORC_TRPL18, ORC_SNGL25, ORC_DBL35, ORC_TRPL52, ORC_UNSUPPORTED = -1 };

int64_t orc_wrap(int64_t v, int w);
int     orc_cmult_regime(int w, int t, int xser);
int     orc_cmult(int64_t d_re, int64_t d_im, int64_t wr, int64_t wi, int w, int t, int xser,
                  int64_t *o_re, int64_t *o_im);
/* twiddle table of one stage: 2^stage entries (rom_twiddle_int + row_twiddle_tay) */
int     orc_twiddles(int stage, int twd, int xser, int64_t *re, int64_t *im);
int     orc_validate(const orc_params *p, int direction);
int     orc_out_width(const orc_params *p, int direction);

void    orc_dif_fly(const orc_params *p, int stage, int dtw, int odd, orc_cplx a, orc_cplx b,
                    int64_t wr, int64_t wi, orc_cplx *x, orc_cplx *y);
void    orc_dit_fly(const orc_params *p, int stage, int dtw, int odd, orc_cplx a, orc_cplx b,
                    int64_t wr, int64_t wi, orc_cplx *x, orc_cplx *y);

/* Core transforms on ONE frame.  "natural" = x[0..N); "bitrev" = v[n] = X[rev(n)], which is
 * the interleaved 2-lane output stream of int_fftNk (lane0[i] = v[2i], lane1[i] = v[2i+1]).
 * stream form: follows math/fn_radix2.m with the RTL butterflies; in-place form: flat array.
 * `p->data_width` is the width at the input of the core being run. */
int orc_fft_stream (const orc_params *p, const orc_cplx *x_nat, orc_cplx *v_bitrev);
int orc_fft_inplace(const orc_params *p, const orc_cplx *x_nat, orc_cplx *v_bitrev);
int orc_ifft_stream (const orc_params *p, const orc_cplx *v_bitrev, orc_cplx *x_nat);
int orc_ifft_inplace(const orc_params *p, const orc_cplx *v_bitrev, orc_cplx *x_nat);

/* index maps of the I/O orders: memory index -> logical (natural) index */
size_t orc_order_index(int order, int log2n, size_t m);

/* Batched driver with the ABI semantics of include/intfft.h: int64 (re,im) interleaved
 * frames, [batch][N][2].  form: 0 stream, 1 in-place.  threads: OpenMP threads (<=0: all). */
int orc_exec(const orc_params *p, int direction, int in_order, int out_order,
             const int64_t *in, int64_t *out, size_t batch, int form, int threads);
/* Same on int16 containers (re,im interleaved), the layout of the headline config. */
int orc_exec_i16(const orc_params *p, int direction, int in_order, int out_order,
                 const int16_t *in, int16_t *out, size_t batch, int form, int threads);
int orc_num_threads(void);

/* N > 512K, "2D-FFT scheme" (int_fftNk.vhd:11-13) -- THIS PROJECT'S EXTENSION, the reference only names it:
 * N = 2^log2n = N1 * N2, N1 = 2^log2_n1 column core first (FWD) / last (INV), both cores native (3..19).
 * Definition and citations: the comment block in intfft_oracle.c.  form: 0 structural with the stream-form
 * cores, 1 flat in-place (what the GPU evaluates), 2 structural with the in-place cores; all three must agree. */
void orc_twiddle_2d(int log2n, int twd, size_t m, int64_t *re, int64_t *im);
int orc_validate_2d(const orc_params *p, int log2_n1, int direction);
int orc_exec_2d(const orc_params *p, int log2_n1, int direction, int in_order, int out_order,
                const int64_t *in, int64_t *out, size_t batch, int form, int threads);

#ifdef __cplusplus
}
#endif
#endif
