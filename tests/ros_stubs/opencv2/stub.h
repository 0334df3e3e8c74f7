#pragma once
#include <cstddef>
#define CV_32FC1 5
namespace cv {
struct Mat {
    int rows, cols; size_t step_; unsigned char *data;
    struct Step { operator size_t() const; } step;
    Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, void *data);
    template <class T> T *ptr(int row = 0);
};
}
