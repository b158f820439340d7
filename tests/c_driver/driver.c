/* driver.c -- the C test driver of INTEGRATION.md section 2, as a program: what a maintainer of the reference would
 * run in place of an RTL simulation of fft_signle_test (src/vhdl/tb/fft_signle_test.vhd:137-166).  Plain C, no Python,
 * no torch: only include/intfft.h and the HIP runtime API.
 *
 *   driver [--sharded rccl|peer] <in.bin> <out.bin> <batch> <nfft> <mode>      mode = 2*FORMAT + RNDMODE (fft_signle_test.vhd:80-112)
 *   --sharded: the batch over every visible HIP device through intfft_exec_sharded (RCCL over xGMI or peer copies)
 * in.bin : batch * 2^nfft (re, im) pairs of int16 (the di_single.dat samples, two integers per line there)
 * out.bin: the natural-order results in the containers intfft_io_widths names (int16 / int32 / int64 pairs)
 * Also exercises the error paths a binding relies on (status codes, not exceptions). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "intfft.h"

/* ---- the function of INTEGRATION.md section 2 ---------------------------------------------------------- */
static int run_single_path(const int16_t *h_frames, void *h_out, size_t batch, int nfft, int mode)
{
    intfft_params p = {.log2n = nfft, .data_width = 16, .twdl_width = 16, .format = mode / 2, .rndmode = mode % 2,
                       .xser = 1, .direction = INTFFT_FWD, .use_fly = 1,
                       .in_order = INTFFT_ORDER_NATURAL, .out_order = INTFFT_ORDER_NATURAL};
    int in_b, out_b, in_c, out_c;
    int rc = intfft_io_widths(&p, &in_b, &out_b, &in_c, &out_c); /* containers: 2/4/8 bytes */
    if (rc) return rc;
    intfft_plan *plan;
    if ((rc = intfft_plan_create(&plan, &p, /*hip_device=*/0))) return rc;
    size_t n = (size_t)1 << nfft;
    void *d_in = NULL, *d_out = NULL;
    if (hipMalloc(&d_in, batch * n * 2 * in_c) != hipSuccess || hipMalloc(&d_out, batch * n * 2 * out_c) != hipSuccess) return 1000;
    if (hipMemcpy(d_in, h_frames, batch * n * 2 * in_c, hipMemcpyHostToDevice) != hipSuccess) return 1001;
    rc = intfft_exec(plan, d_in, d_out, batch, /*stream=*/NULL);
    if (hipMemcpy(h_out, d_out, batch * n * 2 * out_c, hipMemcpyDeviceToHost) != hipSuccess) return 1002; /* syncs */
    /* overlapping, non-identical buffers are refused (include/intfft.h) */
    if (rc == INTFFT_OK && batch > 1 && intfft_exec(plan, d_in, (char *)d_in + 4, batch - 1, NULL) != INTFFT_ERR_INVALID) rc = 1003;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    intfft_plan_destroy(plan);
    return rc;
}

/* ---- driver --sharded: the batch over every visible HIP device ------------------------------------------
 * One plan per device, the frames on device 0; intfft_exec_sharded cuts contiguous shards and moves them with the transport asked for:
 * transport = 1: RCCL (one group of ncclSend / ncclRecv each way, intfft_shard_set_transport), 0: peer copies.  Returns the library's
 * status (INTFFT_ERR_TRANSPORT from the transport call is reported and the run continues on peer copies). */
static int run_sharded(const int16_t *h_frames, void *h_out, size_t batch, int nfft, int mode, int transport, int *used_rccl)
{
    intfft_params p = {.log2n = nfft, .data_width = 16, .twdl_width = 16, .format = mode / 2, .rndmode = mode % 2,
                       .xser = 1, .direction = INTFFT_FWD, .use_fly = 1,
                       .in_order = INTFFT_ORDER_NATURAL, .out_order = INTFFT_ORDER_NATURAL};
    int in_c, out_c, ndev = 0;
    int rc = intfft_io_widths(&p, NULL, NULL, &in_c, &out_c);
    if (rc) return rc;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return INTFFT_ERR_NO_DEVICE;
    if (ndev > 16) ndev = 16;
    intfft_plan *plans[16] = {0};
    for (int d = 0; d < ndev && !rc; ++d) rc = intfft_plan_create(&plans[d], &p, d);
    size_t n = (size_t)1 << nfft;
    void *d_in = NULL, *d_out = NULL;
    if (!rc && (hipSetDevice(0) != hipSuccess || hipMalloc(&d_in, batch * n * 2 * in_c) != hipSuccess ||
                hipMalloc(&d_out, batch * n * 2 * out_c) != hipSuccess ||
                hipMemcpy(d_in, h_frames, batch * n * 2 * in_c, hipMemcpyHostToDevice) != hipSuccess))
        rc = 1000;
    if (!rc) rc = intfft_shard_prepare(plans, ndev, 0, batch);
    *used_rccl = 0;
    if (!rc && transport == INTFFT_TRANSPORT_RCCL) {
        const int rt = intfft_shard_set_transport(plans, ndev, 0, INTFFT_TRANSPORT_RCCL);
        if (rt == INTFFT_OK) *used_rccl = 1;
        else if (rt != INTFFT_ERR_TRANSPORT) rc = rt;
    }
    if (!rc) rc = intfft_exec_sharded(plans, ndev, 0, d_in, d_out, batch); /* blocking */
    if (!rc && hipMemcpy(h_out, d_out, batch * n * 2 * out_c, hipMemcpyDeviceToHost) != hipSuccess) rc = 1002;
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    for (int d = ndev - 1; d >= 0; --d) /* plans[0] owns the communicators: last */
        if (plans[d]) intfft_plan_destroy(plans[d]);
    printf("sharded over %d device(s), transport %s\n", ndev, *used_rccl ? "rccl" : "peer copies");
    return rc;
}

int main(int argc, char **argv)
{
    int sharded = -1; /* --sharded rccl | --sharded peer in front of the positional arguments */
    if (argc == 8 && strcmp(argv[1], "--sharded") == 0) {
        sharded = strcmp(argv[2], "rccl") == 0 ? INTFFT_TRANSPORT_RCCL : INTFFT_TRANSPORT_PEER;
        argv += 2, argc -= 2;
    }
    if (argc != 6) {
        fprintf(stderr, "usage: %s [--sharded rccl|peer] in.bin out.bin batch nfft mode\n", argv[0]);
        return 2;
    }
    const size_t batch = strtoull(argv[3], NULL, 0);
    const int nfft = atoi(argv[4]), mode = atoi(argv[5]);
    const size_t n = (size_t)1 << nfft;
    printf("%s\n", intfft_version());

    /* elaboration failures come back as status codes */
    intfft_params bad = {.log2n = 2, .data_width = 16, .twdl_width = 16, .xser = 1, .use_fly = 1};
    intfft_plan *none = NULL;
    if (intfft_plan_create(&none, &bad, 0) != INTFFT_ERR_INVALID || none) return 3;
    bad.log2n = 10, bad.format = 1, bad.rndmode = 1; /* not elaboratable: int_dif2_fly.vhd:339-346 */
    if (intfft_plan_create(&none, &bad, 0) != INTFFT_ERR_UNSUPPORTED) return 4;
    if (intfft_plan_create(NULL, &bad, 0) != INTFFT_ERR_NULL) return 5;

    intfft_params p = {.log2n = nfft, .data_width = 16, .twdl_width = 16, .format = mode / 2, .rndmode = mode % 2, .xser = 1,
                       .direction = INTFFT_FWD, .use_fly = 1};
    int out_c = 0;
    if (intfft_io_widths(&p, NULL, NULL, NULL, &out_c)) return 6;

    int16_t *in = malloc(batch * n * 2 * sizeof(int16_t));
    void *out = malloc(batch * n * 2 * (size_t)out_c);
    FILE *f = fopen(argv[1], "rb");
    if (!in || !out || !f || fread(in, sizeof(int16_t), batch * n * 2, f) != batch * n * 2) return 7;
    fclose(f);
    int used_rccl = 0;
    const int rc = sharded < 0 ? run_single_path(in, out, batch, nfft, mode) : run_sharded(in, out, batch, nfft, mode, sharded, &used_rccl);
    if (rc) {
        fprintf(stderr, "run_single_path: %s (%d)\n", intfft_strerror(rc), rc);
        return 8;
    }
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, (size_t)out_c, batch * n * 2, f) != batch * n * 2) return 9;
    fclose(f);
    printf("ok batch=%zu nfft=%d mode=%d out_container=%d\n", batch, nfft, mode, out_c);
    free(in);
    free(out);
    return 0;
}
