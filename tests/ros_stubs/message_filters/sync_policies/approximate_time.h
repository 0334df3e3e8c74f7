#pragma once
namespace message_filters { namespace sync_policies { template <class A, class B> struct ApproximateTime { explicit ApproximateTime(unsigned queue); }; } }
