#pragma once
#include <memory>
#include <sensor_msgs/Image.h>
namespace stereo_msgs {
struct DisparityImage { std_msgs::Header header; sensor_msgs::Image image; float f, T, min_disparity, max_disparity, delta_d; };
typedef std::shared_ptr<DisparityImage> DisparityImagePtr;
}
