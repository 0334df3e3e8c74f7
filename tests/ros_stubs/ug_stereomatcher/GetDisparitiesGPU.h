#pragma once
#include <stereo_msgs/DisparityImage.h>
#include <ug_stereomatcher/foveatedstack.h>
// srv/GetDisparitiesGPU.srv
namespace ug_stereomatcher {
struct GetDisparitiesGPU {
    struct Request { sensor_msgs::Image imL, imR; };
    struct Response { stereo_msgs::DisparityImage dispH, dispV, dispC; foveatedstack fdispH, fdispV, fdispC; };
};
}
