// UG_GPU_matcher_ugsm.cpp -- the UG_matcher_gpu node on top of libugsm (catkin build only; not
// compiled in the GPU image, which has no ROS).  Same node / topic / service / parameter names and
// message layouts as /root/reference/src/gpu_matcher/UG_GPU_matcher.cpp (:48-61,:742); the body of
// each callback is: convert to rgb8 -> one call into the MatchGPULib shim -> wrap the returned
// planes in 32FC1 images.  Written from the reference's interface, not from its source text: one
// persistent matcher object instead of one per callback, cv::Mat headers over the returned planes
// instead of per-pixel at<float>() loops, the service path fills the whole fovea stack (U6).
//
// Not in the reference: the parameter `frames_in_flight` (default 1 = the reference's behaviour, one blocking match() per callback,
// UG_GPU_matcher.cpp:423).  With n > 1 the topic path enqueues every synchronised pair into the library's queue (include/ugsm.h:
// ugsm_enqueue_*_managed through the shim's enqueueMatch / enqueueStack) and publishes results as they complete, in arrival order, a
// frame or n - 1 later; the callback blocks only while n pairs are outstanding, and a 1 ms wall timer publishes what finishes between
// frames.  16 MP full mode, one MI355X: 67-70 pairs/s blocking, ~160 pairs/s with frames in flight (bench.py, pcie_inclusive).
#include <cv_bridge/cv_bridge.h>
#include <image_transport/image_transport.h>
#include <image_transport/subscriber_filter.h>
#include <message_filters/sync_policies/approximate_time.h>
#include <message_filters/synchronizer.h>
#include <ros/ros.h>
#include <sensor_msgs/image_encodings.h>
#include <stereo_msgs/DisparityImage.h>
#include <ug_stereomatcher/GetDisparitiesGPU.h>
#include <ug_stereomatcher/foveatedstack.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "MatchGPULib_ugsm.hpp"

namespace enc = sensor_msgs::image_encodings;
using ug_stereomatcher::foveatedstack;

class GPU_matcher {
public:
    GPU_matcher(int argc, char **argv)
        : it_(nh_), imL_sub_(it_, "input_left_image", 1), imR_sub_(it_, "input_right_image", 1),
          sync_(Policy(1), imL_sub_, imR_sub_)
    {
        // frames_in_flight: read once, at start-up (it sizes the context); handed to the shim as "-inflight=N"
        nh_.getParam("frames_in_flight", frames_in_flight_);
        if (frames_in_flight_ < 1) frames_in_flight_ = 1;
        inflight_arg_ = "-inflight=" + std::to_string(frames_in_flight_);
        std::vector<char *> av(argv, argv + argc);
        while (av.size() < 3) av.push_back(const_cast<char *>(av.size() == 2 ? "7" : ""));  // (argv[2] stays the number of fovea levels)
        av.push_back(const_cast<char *>(inflight_arg_.c_str()));
        mgpu_.reset(new MatchGPULib((int)av.size(), av.data()));
        if (frames_in_flight_ > 1) poll_timer_ = nh_.createWallTimer(ros::WallDuration(0.001), &GPU_matcher::pollTimer, this);
        for (const char *t : {"output_stackH", "output_stackV", "output_stackC", "output_stackL_pyramid", "output_stackR_pyramid"})
            stack_pub_[t] = nh_.advertise<foveatedstack>(t, 1);
        for (const char *t : {"output_disparityH", "output_disparityV", "output_disparityC"})
            disp_pub_[t] = nh_.advertise<stereo_msgs::DisparityImage>(t, 1);
        srv_ = nh_.advertiseService("get_disparities_srv", &GPU_matcher::disparitySrv, this);
        sync_.registerCallback(boost::bind(&GPU_matcher::mainRoutine, this, _1, _2));
    }

private:
    typedef message_filters::sync_policies::ApproximateTime<sensor_msgs::Image, sensor_msgs::Image> Policy;
    ros::NodeHandle nh_;
    image_transport::ImageTransport it_;
    image_transport::SubscriberFilter imL_sub_, imR_sub_;
    message_filters::Synchronizer<Policy> sync_;
    ros::ServiceServer srv_;
    std::map<std::string, ros::Publisher> stack_pub_, disp_pub_;
    std::unique_ptr<MatchGPULib> mgpu_;
    int frames_in_flight_ = 1;
    std::string inflight_arg_;
    ros::WallTimer poll_timer_;
    struct InFlight { std_msgs::Header hl, hr; };
    std::map<uint64_t, InFlight> in_flight_;  // headers of the frames whose results have not been published yet
    uint64_t next_tag_ = 0;

    int foveated()
    {  // re-read on every call, default 0 with a warning (reference :96-102)
        int f = 0;
        if (!nh_.getParam("foveated", f)) ROS_WARN("foveated option has not been set. Matcher is on non-foveated mode!");
        return f;
    }
    static sensor_msgs::Image plane_msg(float *p, int rows, int cols, const std_msgs::Header &h)
    {
        cv_bridge::CvImage out(h, enc::TYPE_32FC1, cv::Mat(rows, cols, CV_32FC1, p));
        return *out.toImageMsg();
    }
    // (levels*fovH) x fovW, finest level first
    foveatedstack stack_msg(float ***st, int plane, const std_msgs::Header &h, int imW, int imH, bool dims)
    {
        const int F = mgpu_->getFoveateLevel(), fw = mgpu_->getFoveaWidth(), fh = mgpu_->getFoveaHeight();
        cv::Mat m(F * fh, fw, CV_32FC1);
        for (int k = 0; k < F; k++) std::memcpy(m.ptr<float>(k * fh), st[k][plane], sizeof(float) * fw * fh);
        foveatedstack s;
        s.header = h;
        s.image_stack = *cv_bridge::CvImage(h, enc::TYPE_32FC1, m).toImageMsg();
        if (dims) { s.im_width = imW; s.im_height = imH; s.roi_width = fw; s.roi_height = fh; s.num_levels = F; }
        return s;
    }
    // (levels*fovH) x fovW stack straight from the library's plane (already in the published layout)
    foveatedstack stack_msg(float *plane, int rows_per_level, const std_msgs::Header &h, int imW, int imH)
    {
        const int F = mgpu_->getFoveateLevel(), fw = mgpu_->getFoveaWidth(), fh = mgpu_->getFoveaHeight();
        foveatedstack s;
        s.header = h;
        s.image_stack = plane_msg(plane, F * rows_per_level, fw, h);
        s.im_width = imW; s.im_height = imH; s.roi_width = fw; s.roi_height = fh; s.num_levels = F;
        return s;
    }
    // publishes the oldest finished frame; false if there is none (block == false: or it has not finished)
    bool publishNext(bool block)
    {
        MatchGPULib::Done d;
        if (in_flight_.empty() || !mgpu_->nextDone(block, &d)) return false;
        const InFlight f = in_flight_[d.tag];
        in_flight_.erase(d.tag);
        if (d.status != UGSM_OK) {  // the frame's call failed (reference: exit()): the frame is dropped, the node lives on
            ROS_ERROR("ugsm: frame %llu dropped: %s", (unsigned long long)d.tag, ugsm_status_string(d.status));
            return true;
        }
        if (d.foveated) {
            const int fh = mgpu_->getFoveaHeight();
            if (d.pyramids) {
                stack_pub_["output_stackL_pyramid"].publish(stack_msg(d.planes[3], 3 * fh, f.hl, d.cols, d.rows));
                stack_pub_["output_stackR_pyramid"].publish(stack_msg(d.planes[4], 3 * fh, f.hr, d.cols, d.rows));
            }
            stack_pub_["output_stackH"].publish(stack_msg(d.planes[0], fh, f.hl, d.cols, d.rows));
            stack_pub_["output_stackV"].publish(stack_msg(d.planes[1], fh, f.hr, d.cols, d.rows));
            stack_pub_["output_stackC"].publish(stack_msg(d.planes[2], fh, f.hl, d.cols, d.rows));
        } else {
            const char *topics[3] = {"output_disparityH", "output_disparityV", "output_disparityC"};
            for (int i = 0; i < 3; i++) {
                stereo_msgs::DisparityImage m;
                m.header = (i == 1) ? f.hr : f.hl;
                m.image = plane_msg(d.planes[i], d.rows, d.cols, m.header);
                disp_pub_[topics[i]].publish(m);
            }
        }
        return true;
    }
    void pollTimer(const ros::WallTimerEvent &) { while (publishNext(false)) {} }
    void drain() { while (!in_flight_.empty() && publishNext(true)) {} }

    static void free_stack(float ***st, int F)
    {
        for (int k = 0; k < F; k++) { for (int i = 0; i < 3; i++) free(st[k][i]); free(st[k]); }
        free(st);
    }

    bool disparitySrv(ug_stereomatcher::GetDisparitiesGPU::Request &req, ug_stereomatcher::GetDisparitiesGPU::Response &rsp)
    {
        cv_bridge::CvImagePtr L, R;
        try { L = cv_bridge::toCvCopy(req.imL, enc::RGB8); R = cv_bridge::toCvCopy(req.imR, enc::RGB8); }
        catch (cv_bridge::Exception &) { ROS_ERROR("Could not convert from '%s' to 'rgb8'.", req.imL.encoding.c_str()); return false; }
        const int fov = foveated();
        mgpu_->setFoveated(fov);
        drain();  // (pipelined topic path: the slots belong to the queue while frames are in flight; publish them first)
        if (fov == 1) {
            float ***st = mgpu_->matchStack(L, R);
            if (!st) return false;
            rsp.fdispH = stack_msg(st, 0, L->header, 0, 0, false);
            rsp.fdispV = stack_msg(st, 1, R->header, 0, 0, false);
            rsp.fdispC = stack_msg(st, 2, L->header, 0, 0, false);
            free_stack(st, mgpu_->getFoveateLevel());
        } else {
            float **fin = mgpu_->match(L, R, fov);
            if (!fin) return false;
            rsp.dispH.image = plane_msg(fin[0], L->image.rows, L->image.cols, L->header); rsp.dispH.header = L->header;
            rsp.dispV.image = plane_msg(fin[1], L->image.rows, L->image.cols, R->header); rsp.dispV.header = R->header;
            rsp.dispC.image = plane_msg(fin[2], L->image.rows, L->image.cols, L->header); rsp.dispC.header = L->header;
            for (int i = 0; i < 3; i++) free(fin[i]);
            free(fin);
        }
        return true;
    }

    void mainRoutine(const sensor_msgs::ImageConstPtr &imL, const sensor_msgs::ImageConstPtr &imR)
    {
        cv_bridge::CvImagePtr L, R;
        try { L = cv_bridge::toCvCopy(imL, enc::RGB8); R = cv_bridge::toCvCopy(imR, enc::RGB8); }
        catch (cv_bridge::Exception &) { ROS_ERROR("Could not convert from '%s' to 'rgb8'.", imL->encoding.c_str()); return; }
        const int fov = foveated();
        mgpu_->setFoveated(fov);
        const ros::WallTime t0 = ros::WallTime::now();
        if (frames_in_flight_ > 1) {
            // pipelined: enqueue (the images are copied before the call returns), publish what has finished, block only while
            // frames_in_flight pairs are outstanding
            const uint64_t tag = next_tag_++;
            if (fov == 1) mgpu_->initStack(L, R);
            // (a status other than UGSM_OK = the pair was REJECTED and nothing is outstanding under the tag: include/ugsm.h, ugsm_enqueue_*)
            const int st = fov == 1 ? mgpu_->enqueueStack(L, R, true, tag) : mgpu_->enqueueMatch(L, R, tag);
            if (st != UGSM_OK) { ROS_ERROR("ugsm enqueue failed: %s", ugsm_status_string(st)); return; }
            in_flight_[tag] = InFlight{L->header, R->header};
            while (publishNext(false)) {}
            while (mgpu_->outstanding() >= frames_in_flight_ && publishNext(true)) {}
            return;
        }
        if (fov == 1) {
            mgpu_->initStack(L, R);
            const int F = mgpu_->getFoveateLevel(), fw = mgpu_->getFoveaWidth(), fh = mgpu_->getFoveaHeight();
            // caller-allocated [14][3][fovW*fovH] as the reference node does
            float ***lf = (float ***)malloc(14 * sizeof(float **)), ***rf = (float ***)malloc(14 * sizeof(float **));
            for (int k = 0; k < 14; k++) {
                lf[k] = (float **)malloc(3 * sizeof(float *)); rf[k] = (float **)malloc(3 * sizeof(float *));
                for (int c = 0; c < 3; c++) { lf[k][c] = (float *)malloc(sizeof(float) * fw * fh); rf[k][c] = (float *)malloc(sizeof(float) * fw * fh); }
            }
            float ***st = mgpu_->matchStackPyramid(L, R, lf, rf);
            ROS_INFO("Foveated Disparity took %f Seconds", (ros::WallTime::now() - t0).toSec());
            if (st) {
                auto pyr = [&](float ***p, const std_msgs::Header &h) {  // (F*3*fovH) x fovW, rows [level][channel][row]
                    cv::Mat m(F * 3 * fh, fw, CV_32FC1);
                    for (int k = 0; k < F; k++) for (int c = 0; c < 3; c++) std::memcpy(m.ptr<float>((k * 3 + c) * fh), p[k][c], sizeof(float) * fw * fh);
                    foveatedstack s; s.header = h; s.image_stack = *cv_bridge::CvImage(h, enc::TYPE_32FC1, m).toImageMsg();
                    s.im_width = L->image.cols; s.im_height = L->image.rows; s.roi_width = fw; s.roi_height = fh; s.num_levels = F;
                    return s;
                };
                stack_pub_["output_stackL_pyramid"].publish(pyr(lf, L->header));
                stack_pub_["output_stackR_pyramid"].publish(pyr(rf, R->header));
                stack_pub_["output_stackH"].publish(stack_msg(st, 0, L->header, L->image.cols, L->image.rows, true));
                stack_pub_["output_stackV"].publish(stack_msg(st, 1, R->header, L->image.cols, L->image.rows, true));
                stack_pub_["output_stackC"].publish(stack_msg(st, 2, L->header, L->image.cols, L->image.rows, true));
                free_stack(st, F);
            }
            for (int k = 0; k < 14; k++) { for (int c = 0; c < 3; c++) { free(lf[k][c]); free(rf[k][c]); } free(lf[k]); free(rf[k]); }
            free(lf); free(rf);
        } else {
            float **fin = mgpu_->match(L, R, fov);
            ROS_INFO("Non Foveated Disparity took %f Seconds", (ros::WallTime::now() - t0).toSec());
            if (!fin) return;
            const char *topics[3] = {"output_disparityH", "output_disparityV", "output_disparityC"};
            for (int i = 0; i < 3; i++) {
                stereo_msgs::DisparityImage d;
                d.header = (i == 1) ? R->header : L->header;
                d.image = plane_msg(fin[i], L->image.rows, L->image.cols, d.header);
                disp_pub_[topics[i]].publish(d);
                free(fin[i]);
            }
            free(fin);
        }
    }
};

int main(int argc, char **argv)
{
    ros::init(argc, argv, "RH_GPU_matcher");
    GPU_matcher matcher(argc, argv);
    while (ros::ok()) ros::spin();
    return 0;
}
