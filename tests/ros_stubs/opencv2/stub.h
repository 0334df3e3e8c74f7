#pragma once
#include <cstddef>
#include <ostream>
#include <string>
#include <vector>
#define CV_32FC1 5
#define CV_32F 5
namespace cv {
struct Size { int width, height; Size(); Size(int w, int h); };
std::ostream &operator<<(std::ostream &, const Size &);
struct Mat {
    int rows, cols; size_t step_; unsigned char *data;
    struct Step { operator size_t() const; } step;
    Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, void *data);
    template <class T> T *ptr(int row = 0);
    template <class T> T &at(int row, int col);
    static Mat zeros(int rows, int cols, int type);
    Mat clone() const;
    Size size() const;
};
enum { INTER_CUBIC = 2 };
void resize(const Mat &src, Mat &dst, Size dsize, double fx = 0, double fy = 0, int interpolation = 1);
bool imwrite(const std::string &file, const Mat &img, const std::vector<int> &params = std::vector<int>());
}
