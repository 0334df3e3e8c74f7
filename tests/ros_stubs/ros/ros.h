#pragma once
#include <map>
#include <sstream>
#include <string>
#include <boost/stub.h>
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(args) do { std::ostringstream ros_stub_ss_; ros_stub_ss_ << args; } while (0)
namespace ros {
struct Publisher { template <class M> void publish(const M &) const; };
struct ServiceServer {};
struct WallDuration { WallDuration(); explicit WallDuration(double s); double toSec() const; };
struct WallTime { static WallTime now(); WallDuration operator-(const WallTime &) const; };
struct WallTimerEvent {};
struct WallTimer {};
struct NodeHandle {
    template <class T> WallTimer createWallTimer(WallDuration period, void (T::*)(const WallTimerEvent &), T *obj);
    template <class M> Publisher advertise(const std::string &topic, unsigned queue);
    template <class T, class Req, class Res> ServiceServer advertiseService(const std::string &name, bool (T::*)(Req &, Res &), T *obj);
    bool getParam(const std::string &key, int &v) const;
    bool hasParam(const std::string &key) const;
};
void init(int &argc, char **argv, const std::string &name);
bool ok();
void spin();
}
