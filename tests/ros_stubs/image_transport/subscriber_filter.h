#pragma once
#include <string>
#include <image_transport/image_transport.h>
namespace image_transport { struct SubscriberFilter { SubscriberFilter(ImageTransport &, const std::string &topic, unsigned queue); }; }
