// declarations only (see ../../../README.md)
#pragma once
#include <opencv2/core.hpp>
namespace ORB_SLAM2 { struct MapPoint { cv::Mat GetWorldPos(); }; }
