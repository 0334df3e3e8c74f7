#pragma once
#include <string>
namespace std_msgs { struct String { std::string data; }; }
