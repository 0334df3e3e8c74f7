#pragma once
#include <string>
#include <ros/ros.h>
namespace message_filters { template <class M> struct Subscriber { Subscriber(ros::NodeHandle &, const std::string &, unsigned); }; }
