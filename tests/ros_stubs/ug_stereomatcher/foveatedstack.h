#pragma once
#include <sensor_msgs/Image.h>
// msg/foveatedstack.msg
namespace ug_stereomatcher { struct foveatedstack { std_msgs::Header header; sensor_msgs::Image image_stack; int im_width, im_height, roi_width, roi_height, num_levels; }; }
