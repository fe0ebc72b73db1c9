// declarations only (see ../../../README.md): the members of ORB_SLAM2::KeyFrame the adaptor reads (KeyFrame.h of the reference's fork)
#pragma once
#include <map>
#include <unordered_map>
#include <vector>
#include <opencv2/core.hpp>
#include "MapPoint.h"
namespace ORB_SLAM2 {
struct KeyFrame {
    float fx, fy, cx, cy; int mnMaxX, mnMaxY; long unsigned int mnId;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::map<MapPoint*, int> mmapMpt2Kpt;
    cv::Mat GetPoseSafe() const; cv::Mat GetPoseInverseSafe() const; cv::Mat GetPose() const;
    std::vector<KeyFrame*> GetBestCovisibilityKeyFramesSafe(int n) const; std::vector<KeyFrame*> GetCovisiblesByWeightSafe(int w) const;
    std::unordered_map<int, int> GetUordMatchedKptIds(KeyFrame* other) const;
};
}  // namespace ORB_SLAM2
