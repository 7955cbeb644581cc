// tdsa_spectrum_inst.hip - instantiates the frame kernel for one FFT size (-DTDSA_LOG2N=k), so the
// nine sizes build in parallel.
#include "tdsa_spectrum_kernel.hpp"

#ifndef TDSA_LOG2N
#error "compile with -DTDSA_LOG2N=<6..14>"
#endif

namespace tdsa {
template <>
hipError_t launch_size<TDSA_LOG2N>(int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  return launch_n<TDSA_LOG2N>(in_c64, p, g, s);
}
template <>
LaunchGeom geom_size<TDSA_LOG2N>(int n_frames, int num_cu) {
  return geom_for<TDSA_LOG2N>(n_frames, num_cu);
}
template <>
hipError_t perm_size<TDSA_LOG2N>(const float* w, float* wp, hipStream_t s) {
  return perm_for<TDSA_LOG2N>(w, wp, s);
}
}  // namespace tdsa
