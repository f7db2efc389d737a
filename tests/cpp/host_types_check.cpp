// host_types_check.cpp -- the same lines oracle/ref_src/refcheck.cpp prints from the REAL reference headers, printed
// from the host-side mirrors in include/flucoma_hip/Types.hpp (tests/test_oracle_ref.py compares the two outputs).
#include "../../include/flucoma_hip.h"
#include "../../include/flucoma_hip/Types.hpp"

#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

using fluhip::index;

int main()
{
  std::printf("epsilon %.17g\n", std::numeric_limits<double>::epsilon()); // == fluhip::kEpsilon on the device side
  std::printf("pi %.17g\n", M_PI);
  std::printf("index_bytes %zu signed %d\n", sizeof(index), (int) (index(-1) < 0));
  const index shapes[][2] = {{3, 4}, {862, 1025}, {5168, 1025}, {25840, 2049}, {1, 7}, {7, 1}};
  static double base[1];
  for (auto& sh : shapes)
  {
    fluhip::MatrixView<double> s(base, sh[0], sh[1]);
    fluhip::MatrixView<double> t = s.transpose();
    // the view as the C ABI receives it
    const fluhip_matrix_view cs = s.as<fluhip_matrix_view>(), ct = t.as<fluhip_matrix_view>();
    const bool big = sh[0] > 1 && sh[1] > 2;
    std::printf("slice %ld %ld strides %ld %ld size %ld | transpose extents %ld %ld strides %ld %ld | at(1,2) %ld tat(2,1) %ld\n",
                (long) sh[0], (long) sh[1], (long) cs.row_stride, (long) cs.col_stride, (long) (s.rows() * s.cols()), (long) ct.rows,
                (long) ct.cols, (long) ct.row_stride, (long) ct.col_stride, (long) (big ? &s(1, 2) - base : -1),
                (long) (big ? &t(2, 1) - base : -1));
  }
  {
    std::vector<double> store(17 + 60);
    fluhip::MatrixView<double> s(store.data() + 17, 10, 6);
    std::printf("offset_slice start %ld at(0,0) %ld at(9,5) %ld\n", (long) (s.data() - store.data()), (long) (&s(0, 0) - s.data()),
                (long) (&s(9, 5) - s.data()));
  }
  {
    fluhip::FluidTask task;
    const double seq[][4] = {{0, 2, 1, 200}, {0, 2, 100, 200}, {1, 2, 50, 200}, {1, 2, 200, 200}, {0, 1, 7, 50}, {2, 3, 0, 10}};
    for (auto& q : seq)
    {
      const bool a = task.iterationUpdate(q[0], q[1]);
      const bool b = task.processUpdate(q[2], q[3]);
      std::printf("task iter %g/%g done %g/%g -> %d %d progress %.17g\n", q[0], q[1], q[2], q[3], (int) a, (int) b, task.progress());
    }
    task.cancel();
    std::printf("task cancelled %d update %d iteration %d\n", (int) task.cancelled(), (int) task.processUpdate(1, 2), (int) task.iterationUpdate(0, 1));
    task.reset();
    std::printf("task reset %d update %d\n", (int) task.cancelled(), (int) task.processUpdate(1, 2));
  }
  {
    using fluhip::Result;
    Result ok;
    Result err{Result::Status::kError, "Input buffer ", "x", ": not enough frames"};
    Result warn{Result::Status::kWarning, "w"};
    Result canc{Result::Status::kCancelled, ""};
    std::printf("result ok %d %d '%s'\n", (int) ok.ok(), (int) ok.status(), ok.message().c_str());
    std::printf("result err %d %d '%s'\n", (int) err.ok(), (int) err.status(), err.message().c_str());
    std::printf("result warn %d %d '%s'\n", (int) warn.ok(), (int) warn.status(), warn.message().c_str());
    std::printf("result cancelled %d %d '%s'\n", (int) canc.ok(), (int) canc.status(), canc.message().c_str());
    err.addMessage(" more ", 3);
    std::printf("result added '%s'\n", err.message().c_str());
    err.set(Result::Status::kWarning);
    std::printf("result set %d\n", (int) err.status());
    // the C ABI's status codes are the same integers (include/flucoma_hip.h fluhip_status)
    std::printf("abi_status %d %d %d %d\n", FLUHIP_OK, FLUHIP_WARNING, FLUHIP_ERROR, FLUHIP_CANCELLED);
  }
  return 0;
}
