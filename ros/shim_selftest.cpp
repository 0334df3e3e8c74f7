// Compiles the MatchGPULib shim without OpenCV/ROS (a struct with the same field names stands in
// for cv_bridge::CvImagePtr) and runs one small full-mode and one foveated match through it.
//   g++ -std=c++17 -Iinclude ros/shim_selftest.cpp -Lug_stereomatcher_amd -lugsm -Wl,-rpath,$PWD/ug_stereomatcher_amd
#include <cstdint>
#include <memory>
#include <vector>

#include "MatchGPULib_ugsm.hpp"

struct Mat { int rows, cols; size_t step; unsigned char *data; };
struct CvImage { Mat image; };
typedef std::shared_ptr<CvImage> CvImagePtr;

int main(int argc, char **argv)
{
    const int W = 320, H = 240;
    std::vector<uint8_t> l(3 * W * H), r(3 * W * H);
    unsigned s = 1;
    for (size_t i = 0; i < l.size(); i++) { s = s * 1664525u + 1013904223u; l[i] = 1 + (s >> 24) % 255; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) r[(y * W + x) * 3 + c] = l[(y * W + (x >= 2 ? x - 2 : 0)) * 3 + c];
    CvImagePtr L(new CvImage{{H, W, (size_t)3 * W, l.data()}}), R(new CvImage{{H, W, (size_t)3 * W, r.data()}});
    char *av[] = {(char *)"node", (char *)"x", (char *)"3"};
    try {
        MatchGPULib m(3, av);
        m.initStack(L, R);
        std::printf("fovea %dx%d levels %d\n", m.getFoveaWidth(), m.getFoveaHeight(), m.getFoveateLevel());
        float ***st = m.matchStack(L, R);
        if (!st) return 2;
        std::printf("stack[0][0][centre] = %f\n", st[0][0][(m.getFoveaHeight() / 2) * m.getFoveaWidth() + m.getFoveaWidth() / 2]);
        // match(L, R, 1) == matchStack + hierarchicalDisparity (MatchGPULib.cpp:354-360)
        float **full = m.hierarchicalDisparity(nullptr, st, 3, W, H);
        float **one = m.match(L, R, 1);
        if (!full || !one) return 3;
        size_t diff = 0;
        for (int c = 0; c < 3; c++) diff += std::memcmp(full[c], one[c], sizeof(float) * W * H) != 0;
        std::printf("hierarchicalDisparity vs match(fov=1): %s, dx[centre] = %f, dx[corner] = %f\n", diff ? "DIFFER" : "identical",
                    full[0][(H / 2) * W + W / 2], full[0][0]);
        for (int c = 0; c < 3; c++) { free(full[c]); free(one[c]); }
        free(full);
        free(one);
        // the pipelined calls ("-inflight=3"): five frames enqueued, results in arrival order, equal to the blocking calls
        {
            char *av2[] = {(char *)"node", (char *)"x", (char *)"3", (char *)"-inflight=3"};
            MatchGPULib p(4, av2);
            float **blocking = p.match(L, R, 0);
            if (!blocking) return 4;
            size_t bad = 0, got = 0;
            uint64_t expect_tag = 0;
            MatchGPULib::Done d;
            auto check = [&](const MatchGPULib::Done &dn) {
                bad += dn.tag != expect_tag++;
                if (!dn.foveated) for (int c = 0; c < 3; c++) bad += std::memcmp(dn.planes[c], blocking[c], sizeof(float) * W * H) != 0;
                else for (int k = 0; k < p.getFoveateLevel(); k++)
                    for (int c = 0; c < 3; c++)
                        bad += std::memcmp(dn.planes[c] + (size_t)k * p.getFoveaWidth() * p.getFoveaHeight(), st[k][c], sizeof(float) * p.getFoveaWidth() * p.getFoveaHeight()) != 0;
                got++;
            };
            for (uint64_t t = 0; t < 5; t++) {
                if ((t & 1 ? p.enqueueStack(L, R, false, t) : p.enqueueMatch(L, R, t)) != UGSM_OK) return 5;
                while (p.outstanding() >= p.frames_in_flight) { if (!p.nextDone(true, &d)) return 6; check(d); }
            }
            while (p.outstanding() > 0) { if (!p.nextDone(true, &d)) return 7; check(d); }
            std::printf("pipelined (3 in flight) vs blocking: %zu frames, %s\n", got, bad ? "DIFFER" : "identical");
            for (int c = 0; c < 3; c++) free(blocking[c]);
            free(blocking);
            if (bad || got != 5) return 8;
        }
        for (int k = 0; k < m.getFoveateLevel(); k++) { for (int i = 0; i < 3; i++) free(st[k][i]); free(st[k]); }
        free(st);
    } catch (const std::exception &e) {
        std::printf("no device: %s\n", e.what());
        return 0;
    }
    return 0;
}
