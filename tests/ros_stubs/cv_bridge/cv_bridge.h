#pragma once
#include <exception>
#include <memory>
#include <string>
#include <opencv2/stub.h>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
struct Exception : std::exception {};
struct CvImage {
    std_msgs::Header header; std::string encoding; cv::Mat image;
    CvImage(); CvImage(const std_msgs::Header &, const std::string &, const cv::Mat &);
    sensor_msgs::ImagePtr toImageMsg() const;
};
typedef std::shared_ptr<CvImage> CvImagePtr;
CvImagePtr toCvCopy(const sensor_msgs::ImageConstPtr &, const std::string &);
CvImagePtr toCvCopy(const sensor_msgs::Image &, const std::string &);
}
