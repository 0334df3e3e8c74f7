#pragma once
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
struct arg1 {}; struct arg2 {};
template <class F, class C, class A, class B> int bind(F, C, A, B);
}
static boost::arg1 _1; static boost::arg2 _2;
