#pragma once
namespace message_filters {
template <class Policy> struct Synchronizer {
    template <class F0, class F1> Synchronizer(const Policy &, F0 &, F1 &);
    template <class C> void registerCallback(const C &);
};
}
