// MatchGPULib_ugsm.hpp -- source-compatible C++ shim of the reference's MatchGPULib class
// (/root/reference/src/gpu_matcher/MatchGPULib.h:6-47) over the C-ABI of include/ugsm.h.
//
// The node's call sites (UG_GPU_matcher.cpp:160-181,423,530-535,645) compile unchanged against
// this header: same constructor, same method names, same return layouts (malloc'd float** /
// float***, freed by the caller exactly as the node already does, :414-418,487-489,636-640,689-691).
// The image argument is templated on "something with ->image.{rows,cols,step,data}", i.e.
// cv_bridge::CvImagePtr in the node; the header itself needs neither OpenCV nor ROS.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "ugsm.h"

class MatchGPULib {
public:
    bool foveatedmatching;
    int foveatelevel;
    int fovH;
    int fovW;
    int frames_in_flight = 1;  // (not in the reference) pairs the pipelined calls keep outstanding; "-inflight=N"

    // MatchGPULib.cpp:251-265: "-device=N" anywhere in argv, argv[2] = number of fovea levels (7)
    MatchGPULib(int argc, char **argv) : foveatedmatching(false), foveatelevel(7), fovH(0), fovW(0), ctx_(nullptr)
    {
        ugsm_config cfg;
        ugsm_default_config(&cfg);
        for (int i = 1; i < argc; i++) {
            if (argv[i] && std::strncmp(argv[i], "-device=", 8) == 0) cfg.device = std::atoi(argv[i] + 8);
            // not in the reference: "-lrcheck=TAU" switches the optional LR-consistency check on (ugsm_config.lr_check_threshold;
            // full mode only; off by default, and the results are the reference's only while it is off)
            if (argv[i] && std::strncmp(argv[i], "-lrcheck=", 9) == 0) cfg.lr_check_threshold = (float)std::atof(argv[i] + 9);
            // not in the reference: "-inflight=N" sizes the context for the pipelined calls below (N pairs outstanding in the library's
            // queue: min(N, 4) slots, ceil(N / slots) pairs per call at most); 1 = the reference's one blocking call at a time
            if (argv[i] && std::strncmp(argv[i], "-inflight=", 10) == 0) frames_in_flight = std::atoi(argv[i] + 10);
        }
        if (frames_in_flight < 1) frames_in_flight = 1;
        cfg.slots = frames_in_flight < 4 ? frames_in_flight : 4;
        cfg.batch = (frames_in_flight + cfg.slots - 1) / cfg.slots;
        if (cfg.batch > 8) cfg.batch = 8;
        if (argc > 2) foveatelevel = std::atoi(argv[2]);
        cfg.fovea_levels = foveatelevel;
        const int st = ugsm_create(&cfg, &ctx_);
        if (st != UGSM_OK) throw std::runtime_error(std::string("ugsm_create: ") + ugsm_status_string(st));
    }
    ~MatchGPULib() { ugsm_destroy(ctx_); }
    MatchGPULib(const MatchGPULib &) = delete;
    MatchGPULib &operator=(const MatchGPULib &) = delete;

    int getFoveaWidth() { return fovW; }
    int getFoveaHeight() { return fovH; }
    int getFoveateLevel() { return foveatelevel; }
    void setFoveaWidth(int rows) { fovW = rows; }
    void setFoveaHeight(int cols) { fovH = cols; }
    void setFoveated(int fov) { foveatedmatching = fov; }

    // MatchGPULib.cpp:406-426
    template <class ImgPtr>
    int initStack(ImgPtr L, ImgPtr /*R*/)
    {
        return ugsm_fovea_dims(L->image.cols, L->image.rows, 14, foveatelevel, &fovW, &fovH) == UGSM_OK ? 0 : -1;
    }

    // MatchGPULib.cpp:303-403 (fov == 0).  Returns finDisp[3][rows*cols]; nullptr on failure
    // (the reference exit()s instead).
    template <class ImgPtr>
    float **match(ImgPtr L, ImgPtr R, int fov)
    {
        foveatedmatching = fov;
        const int W = L->image.cols, H = L->image.rows;
        if (R->image.cols != W || R->image.rows != H) return nullptr;
        float **fin = alloc_planes(3, (size_t)W * H);
        // fov == 1: foveated matching + hierarchicalDisparity (MatchGPULib.cpp:354-360)
        const int st = fov == 1 ? ugsm_match_foveated_full(ctx_, L->image.data, R->image.data, W, H, (int)L->image.step, 0, 0, fin[0], fin[1], fin[2])
                                : ugsm_match_full(ctx_, L->image.data, R->image.data, W, H, (int)L->image.step, fin[0], fin[1], fin[2]);
        if (st != UGSM_OK) return fail(fin, 3, st);
        return fin;
    }

    // MatchGPULib.cpp:429-531.  Returns disparity[level][3][fovH*fovW] for level < foveatelevel.
    template <class ImgPtr>
    float ***matchStack(ImgPtr L, ImgPtr R) { return stack(L, R, nullptr, nullptr); }

    // MatchGPULib.cpp:534-700.  leftFov/rightFov: caller-allocated [14][3][fovH*fovW] as in
    // UG_GPU_matcher.cpp:169-179; levels < foveatelevel are filled.
    template <class ImgPtr>
    float ***matchStackPyramid(ImgPtr L, ImgPtr R, float ***leftFov, float ***rightFov) { return stack(L, R, leftFov, rightFov); }

    // MatchGPULib.cpp:2589-2701.  foveated[level][channel] = fovW*fovH floats for level < foveatelevel (what
    // matchStack returns); `im` is unused, as in the reference.  Returns 3 malloc'd planes widthInit*heightInit.
    float **hierarchicalDisparity(float ** /*im*/, float ***foveated, int channels, int widthInit, int heightInit)
    {
        if (channels != 3 || !foveated) return nullptr;
        const int F = foveatelevel;
        if (ugsm_fovea_dims(widthInit, heightInit, 14, F, &fovW, &fovH) != UGSM_OK) return nullptr;
        const size_t fn = (size_t)fovW * fovH, sn = fn * F, n = (size_t)widthInit * heightInit;
        void *d_stack = nullptr, *d_out = nullptr;
        float **fin = alloc_planes(3, n);
        int st = ugsm_dev_alloc(ctx_, &d_stack, (long long)(3 * sn * sizeof(float)));
        if (st == UGSM_OK) st = ugsm_dev_alloc(ctx_, &d_out, (long long)(3 * n * sizeof(float)));
        for (int c = 0; c < 3 && st == UGSM_OK; c++)
            for (int k = 0; k < F && st == UGSM_OK; k++)
                st = ugsm_copy_to_device(ctx_, (float *)d_stack + c * sn + k * fn, foveated[k][c], (long long)(fn * sizeof(float)));
        if (st == UGSM_OK)
            st = ugsm_reconstruct_full(ctx_, 0, (float *)d_stack, (float *)d_stack + sn, (float *)d_stack + 2 * sn, widthInit, heightInit, 0, 0,
                                       (float *)d_out);
        if (st == UGSM_OK) st = ugsm_wait(ctx_, 0);
        for (int c = 0; c < 3 && st == UGSM_OK; c++) st = ugsm_copy_to_host(ctx_, fin[c], (float *)d_out + c * n, (long long)(n * sizeof(float)));
        if (d_stack) ugsm_dev_free(ctx_, d_stack);
        if (d_out) ugsm_dev_free(ctx_, d_out);
        if (st != UGSM_OK) return fail(fin, 3, st);
        return fin;
    }

    // ---- the pipelined twin of match / matchStack / matchStackPyramid (not in the reference): include/ugsm.h, "the queue" ------------
    // The images are copied into page-locked staging memory of the library before the call returns (the cv_bridge image may be freed);
    // the result comes out of nextDone, in arrival order.  Every call flushes: a frame starts at once if a slot is free, and under load
    // the backlog batches itself.
    template <class ImgPtr>
    int enqueueMatch(ImgPtr L, ImgPtr R, uint64_t tag)
    {
        const int W = L->image.cols, H = L->image.rows;
        if (R->image.cols != W || R->image.rows != H) return UGSM_ERR_SIZE_MISMATCH;
        int st = ugsm_enqueue_full_managed(ctx_, L->image.data, R->image.data, W, H, (int)L->image.step, tag);
        if (st == UGSM_OK) { kinds_[tag] = Kind{false, false, W, H}; st = ugsm_flush(ctx_); }
        return report(st);
    }
    template <class ImgPtr>
    int enqueueStack(ImgPtr L, ImgPtr R, bool want_pyramids, uint64_t tag)
    {
        const int W = L->image.cols, H = L->image.rows;
        if (R->image.cols != W || R->image.rows != H) return UGSM_ERR_SIZE_MISMATCH;
        if (ugsm_fovea_dims(W, H, 14, foveatelevel, &fovW, &fovH) != UGSM_OK) return UGSM_ERR_BAD_ARG;
        int st = ugsm_enqueue_foveated_managed(ctx_, L->image.data, R->image.data, W, H, (int)L->image.step, 0, 0, want_pyramids ? 1 : 0, tag);
        if (st == UGSM_OK) { kinds_[tag] = Kind{true, want_pyramids, W, H}; st = ugsm_flush(ctx_); }
        return report(st);
    }
    int outstanding() const { return (int)kinds_.size(); }
    // What nextDone hands out.  planes: full mode dispH, dispV, dispC (rows x cols); foveated stackH, stackV, stackC ((levels fovH) x fovW,
    // finest level first -- the layout the node publishes, UG_GPU_matcher.cpp:293-320) and, if asked for, the L / R pyramid stacks
    // ((levels 3 fovH) x fovW, :203-226).  The planes belong to the library and stay valid until the next nextDone.
    struct Done {
        uint64_t tag;
        int status;      // UGSM_OK, or the status of the library call the pair went out in (planes are then null: nothing to publish)
        bool foveated, pyramids;
        int rows, cols;  // of the image
        float *planes[5];
    };
    // true: *out filled -- the oldest pair has been REPORTED and no longer counts as outstanding; out->status says how its call went (a
    // failed call: logged to stderr, planes null; the caller drops whatever it keeps under the tag).  false: nothing outstanding, or
    // (block == false) the oldest pair has not finished.
    bool nextDone(bool block, Done *out)
    {
        ugsm_completion c;
        const int st = ugsm_next_done(ctx_, &c, block ? 1 : 0);
        if (st != UGSM_OK) { if (st != UGSM_PENDING && st != UGSM_EMPTY) report(st); return false; }
        const Kind k = kinds_[c.tag];
        kinds_.erase(c.tag);
        out->tag = c.tag; out->status = report(c.status); out->foveated = k.foveated; out->pyramids = k.pyramids; out->rows = k.H; out->cols = k.W;
        for (int i = 0; i < 5; i++) out->planes[i] = c.status == UGSM_OK ? c.result[i] : nullptr;
        return true;
    }

private:
    ugsm_ctx *ctx_;
    struct Kind { bool foveated, pyramids; int W, H; };
    std::map<uint64_t, Kind> kinds_;
    int report(int st)
    {
        if (st != UGSM_OK) std::fprintf(stderr, "ugsm: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx_));
        return st;
    }

    static float **alloc_planes(int n, size_t px)
    {
        float **p = (float **)std::malloc(n * sizeof(float *));
        for (int i = 0; i < n; i++) p[i] = (float *)std::malloc(px * sizeof(float));
        return p;
    }
    float **fail(float **p, int n, int st)
    {
        std::fprintf(stderr, "ugsm: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx_));
        for (int i = 0; i < n; i++) std::free(p[i]);
        std::free(p);
        return nullptr;
    }
    template <class ImgPtr>
    float ***stack(ImgPtr L, ImgPtr R, float ***leftFov, float ***rightFov)
    {
        const int W = L->image.cols, H = L->image.rows, F = foveatelevel;
        if (R->image.cols != W || R->image.rows != H) return nullptr;
        if (ugsm_fovea_dims(W, H, 14, F, &fovW, &fovH) != UGSM_OK) return nullptr;
        const size_t fn = (size_t)fovW * fovH, sn = fn * F;
        float *sh = (float *)std::malloc(3 * sn * sizeof(float));
        float *pl = leftFov ? (float *)std::malloc(3 * sn * sizeof(float)) : nullptr;
        float *pr = rightFov ? (float *)std::malloc(3 * sn * sizeof(float)) : nullptr;
        const int st = ugsm_match_foveated(ctx_, L->image.data, R->image.data, W, H, (int)L->image.step, 0, 0, sh, sh + sn,
                                           sh + 2 * sn, pl, pr);
        float ***disp = nullptr;
        if (st == UGSM_OK) {
            disp = (float ***)std::malloc(F * sizeof(float **));
            for (int k = 0; k < F; k++) {
                disp[k] = alloc_planes(3, fn);
                for (int c = 0; c < 3; c++) std::memcpy(disp[k][c], sh + c * sn + k * fn, fn * sizeof(float));
                for (int c = 0; c < 3 && pl; c++) std::memcpy(leftFov[k][c], pl + ((size_t)k * 3 + c) * fn, fn * sizeof(float));
                for (int c = 0; c < 3 && pr; c++) std::memcpy(rightFov[k][c], pr + ((size_t)k * 3 + c) * fn, fn * sizeof(float));
            }
        } else {
            std::fprintf(stderr, "ugsm: %s: %s\n", ugsm_status_string(st), ugsm_last_error(ctx_));
        }
        std::free(sh);
        std::free(pl);
        std::free(pr);
        return disp;
    }
};
