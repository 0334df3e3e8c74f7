#pragma once
#include <string>
namespace ros { namespace package { std::string getPath(const std::string &); } }
