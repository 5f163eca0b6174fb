// test double, see core.hpp
#pragma once
#include "core.hpp"
namespace cv { Mat imread(const std::string &filename, int flags = 1); }
