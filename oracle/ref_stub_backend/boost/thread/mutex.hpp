// TEST INFRASTRUCTURE: stand-in for <boost/thread/mutex.hpp> (Boost is not in this image) over the C++11 primitives the
// Boost ones were standardised from -- what tandem_backend.{h,cpp} uses: boost::mutex, boost::unique_lock<boost::mutex>,
// boost::condition_variable (wait(lock), notify_all()).
#pragma once
#include <condition_variable>
#include <mutex>

namespace boost {
using mutex = std::mutex;
template <class M> using unique_lock = std::unique_lock<M>;
using condition_variable = std::condition_variable;
}  // namespace boost
