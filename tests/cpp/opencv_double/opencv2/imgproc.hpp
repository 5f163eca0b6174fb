// test double, see core.hpp
#pragma once
#include "core.hpp"
namespace cv { enum { COLOR_BGR2GRAY = 6 }; void cvtColor(const Mat &src, Mat &dst, int code, int dstCn = 0); }
