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
        for (int k = 0; k < m.getFoveateLevel(); k++) { for (int i = 0; i < 3; i++) free(st[k][i]); free(st[k]); }
        free(st);
    } catch (const std::exception &e) {
        std::printf("no device: %s\n", e.what());
        return 0;
    }
    return 0;
}
