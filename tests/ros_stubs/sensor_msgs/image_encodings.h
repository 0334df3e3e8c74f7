#pragma once
#include <string>
namespace sensor_msgs { namespace image_encodings { extern const std::string RGB8; extern const std::string TYPE_32FC1; } }
