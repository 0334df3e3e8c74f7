/* queue_example.c -- the queue of include/ugsm.h from PLAIN C (no C++, no Python, no ROS): what a host's frame loop looks like.
 *
 *   gcc -std=c99 -O1 -Iinclude ros/queue_example.c -Lug_stereomatcher_amd -lugsm -Wl,-rpath,$PWD/ug_stereomatcher_amd -o queue_example
 *
 * Seven synthetic frames go through ugsm_enqueue_full_managed (any host memory in, library-owned planes out) with three frames in
 * flight; every result is compared with the blocking ugsm_match_full of the same frame (the reference node's call,
 * UG_GPU_matcher.cpp:423), byte for byte.  Prints "QUEUE_EXAMPLE_OK" on success; tests/test_gpu_queue.py runs it on the GPU box. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ugsm.h"

#define W 320
#define H 240
#define FRAMES 7
#define IN_FLIGHT 3

static void make_frame(int k, uint8_t *l, uint8_t *r)
{
    unsigned s = 12345u + 977u * (unsigned)k;
    for (size_t i = 0; i < (size_t)3 * W * H; i++) {
        s = s * 1664525u + 1013904223u;
        l[i] = (uint8_t)(1 + (s >> 24) % 255);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) r[(y * W + x) * 3 + c] = l[(y * W + (x >= 2 + k % 3 ? x - 2 - k % 3 : 0)) * 3 + c];
}

int main(void)
{
    ugsm_config cfg;
    ugsm_ctx *ctx = NULL;
    ugsm_default_config(&cfg);
    cfg.slots = 2;
    cfg.batch = 2;
    int st = ugsm_create(&cfg, &ctx);
    if (st != UGSM_OK) {
        printf("no device: %s\n", ugsm_status_string(st));
        return 0;
    }
    const size_t px = (size_t)W * H;
    uint8_t *l = malloc(3 * px), *r = malloc(3 * px);
    float *expect = malloc(FRAMES * 3 * px * sizeof(float));
    for (int k = 0; k < FRAMES; k++) {  /* the blocking call, frame by frame */
        make_frame(k, l, r);
        float *e = expect + (size_t)k * 3 * px;
        st = ugsm_match_full(ctx, l, r, W, H, 3 * W, e, e + px, e + 2 * px);
        if (st != UGSM_OK) { printf("ugsm_match_full: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx)); return 1; }
    }
    int outstanding = 0, reported = 0, bad = 0, calls_seen = 0;
    long long last_call = -1;
    ugsm_completion c;
    for (int k = 0; k < FRAMES || outstanding > 0;) {
        if (k < FRAMES && outstanding < IN_FLIGHT) {  /* a frame arrives */
            make_frame(k, l, r);
            st = ugsm_enqueue_full_managed(ctx, l, r, W, H, 3 * W, (uint64_t)k);
            memset(l, 0, 3 * px);                      /* the images were copied before the call returned: the buffers are ours again */
            memset(r, 0, 3 * px);
            if (st == UGSM_OK) st = ugsm_flush(ctx);
            if (st != UGSM_OK) { printf("enqueue: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx)); return 1; }
            k++;
            outstanding++;
        }
        /* publish what has finished; block only when the pipe is full (or nothing is left to enqueue) */
        const int block = (outstanding >= IN_FLIGHT || k >= FRAMES) ? 1 : 0;
        while ((st = ugsm_next_done(ctx, &c, block && outstanding > 0)) == UGSM_OK) {
            const float *e = expect + (size_t)c.tag * 3 * px;
            if (c.status != UGSM_OK || (int)c.tag != reported) bad++;
            for (int p = 0; p < 3; p++) bad += memcmp(c.result[p], e + (size_t)p * px, px * sizeof(float)) != 0;
            if (c.call_index != last_call) { calls_seen++; last_call = c.call_index; }
            reported++;
            outstanding--;
            if (block) break;
        }
        if (st != UGSM_OK && st != UGSM_PENDING && st != UGSM_EMPTY) { printf("next_done: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx)); return 1; }
    }
    printf("%d frames, %d in flight, %d library calls, %lld bytes of device memory: %s\n", reported, IN_FLIGHT, calls_seen, ugsm_context_device_bytes(ctx),
           bad ? "DIFFER" : "identical to the blocking calls");
    free(l);
    free(r);
    free(expect);
    ugsm_destroy(ctx);
    if (bad || reported != FRAMES) return 2;
    printf("QUEUE_EXAMPLE_OK\n");
    return 0;
}
