#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Image { std_msgs::Header header; uint32_t height, width; std::string encoding; uint8_t is_bigendian; uint32_t step; std::vector<uint8_t> data; };
typedef std::shared_ptr<Image const> ImageConstPtr;
typedef std::shared_ptr<Image> ImagePtr;
}
