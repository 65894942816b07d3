// TEST INFRASTRUCTURE: stand-in for <boost/thread/thread.hpp>.  boost::thread as tandem_backend.cpp uses it (:21: move-assigned
// from `boost::thread(&TandemBackendImpl::Loop, this)`, never joined): a std::thread that DETACHES in its destructor, which is
// what Boost.Thread's default (BOOST_THREAD_VERSION 2) does -- std::thread would call std::terminate().
#pragma once
#include <thread>
#include <utility>

#include "mutex.hpp"

namespace boost {
class thread {
public:
  thread() = default;
  template <class F, class... A> explicit thread(F &&f, A &&...a) : t_(std::forward<F>(f), std::forward<A>(a)...) {}
  thread(thread &&o) noexcept : t_(std::move(o.t_)) {}
  thread &operator=(thread &&o) noexcept {
    if (t_.joinable()) t_.detach();
    t_ = std::move(o.t_);
    return *this;
  }
  ~thread() { if (t_.joinable()) t_.detach(); }
  void join() { t_.join(); }
  bool joinable() const { return t_.joinable(); }
private:
  std::thread t_;
};
}  // namespace boost
