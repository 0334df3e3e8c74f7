#pragma once
#include <ros/ros.h>
namespace image_transport { struct ImageTransport { explicit ImageTransport(const ros::NodeHandle &); }; }
