// refcheck.cpp -- the part of the REAL reference that can be compiled in this image: the headers on the BufNMF path
// that need neither Eigen, HISSTools nor foonathan-memory (everything else on the path includes one of them, see
// DESIGN section 5).  Built by oracle/Makefile from the sources where they lie under /root/reference into
// oracle/_ref/refcheck; prints known answers that tests/test_oracle_ref.py compares with the host-side mirrors in
// include/flucoma_hip/Types.hpp (tests/cpp/host_types_check.cpp prints the same lines from the mirror).
//   flucoma/data/FluidIndex.hpp, flucoma/data/FluidTensor_Support.hpp   (strides, transpose, slices)
//   flucoma/clients/common/FluidTask.hpp                                (a12: progress arithmetic, cancellation)
//   flucoma/clients/common/Result.hpp                                   ((b): status / message convention)
//   flucoma/algorithms/util/AlgorithmUtils.hpp                          (epsilon, pi)
#include <flucoma/algorithms/util/AlgorithmUtils.hpp>
#include <flucoma/clients/common/FluidTask.hpp>
#include <flucoma/clients/common/Result.hpp>
#include <flucoma/data/FluidIndex.hpp>
#include <flucoma/data/FluidTensor_Support.hpp>

#include <cstdio>

using fluid::FluidTensorSlice;
using fluid::index;

int main()
{
  std::printf("epsilon %.17g\n", fluid::algorithm::epsilon);
  std::printf("pi %.17g\n", fluid::algorithm::pi);
  std::printf("index_bytes %zu signed %d\n", sizeof(index), (int) (index(-1) < 0));
  // row-major strides of a T x F matrix, its transpose() and element offsets (what asEigen maps)
  const index shapes[][2] = {{3, 4}, {862, 1025}, {5168, 1025}, {25840, 2049}, {1, 7}, {7, 1}};
  for (auto& sh : shapes)
  {
    FluidTensorSlice<2> s(0, {sh[0], sh[1]});
    FluidTensorSlice<2> t = s.transpose();
    std::printf("slice %ld %ld strides %ld %ld size %ld | transpose extents %ld %ld strides %ld %ld | at(1,2) %ld tat(2,1) %ld\n",
                (long) sh[0], (long) sh[1], (long) s.strides[0], (long) s.strides[1], (long) s.size, (long) t.extents[0],
                (long) t.extents[1], (long) t.strides[0], (long) t.strides[1],
                (long) (sh[0] > 1 && sh[1] > 2 ? s(index(1), index(2)) : -1), (long) (sh[0] > 1 && sh[1] > 2 ? t(index(2), index(1)) : -1));
  }
  // a slice with a start offset (FluidTensorView::data() = ref + start)
  {
    FluidTensorSlice<2> s(17, {10, 6});
    std::printf("offset_slice start %ld at(0,0) %ld at(9,5) %ld\n", (long) s.start, (long) s(index(0), index(0)), (long) s(index(9), index(5)));
  }
  // FluidTask: progress = done / (total * nIterTotal) + iter / nIterTotal (clients/common/FluidTask.hpp:22-34)
  {
    fluid::FluidTask task;
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
  // Result: status codes, ok(), message concatenation
  {
    using fluid::client::Result;
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
  }
  return 0;
}
