#pragma once
#include <std_msgs/Header.h>
namespace sensor_msgs { struct JointState { std_msgs::Header header; }; }
