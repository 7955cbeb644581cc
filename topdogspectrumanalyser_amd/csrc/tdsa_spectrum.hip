// tdsa_spectrum.hip - size dispatch for the fused frame kernel (kernels live in
// tdsa_spectrum_kernel.hpp, one instantiation TU per size).
#include "tdsa_kernels.hpp"

namespace tdsa {
template <int LOG2N> hipError_t launch_size(int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s);
template <int LOG2N> LaunchGeom geom_size(int n_frames, int num_cu);
template <int LOG2N> hipError_t perm_size(const float* w, float* wp, hipStream_t s);

#define TDSA_FOR_SIZES(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)
#define TDSA_DECL(k)                                                                                  \
  template <> hipError_t launch_size<k>(int, const SpecParams&, const LaunchGeom&, hipStream_t);      \
  template <> LaunchGeom geom_size<k>(int, int);                                                      \
  template <> hipError_t perm_size<k>(const float*, float*, hipStream_t);
TDSA_FOR_SIZES(TDSA_DECL)

LaunchGeom spectrum_geometry(int log2n, int n_frames, int num_cu) {
  switch (log2n) {
#define TDSA_CASE(k) case k: return geom_size<k>(n_frames, num_cu);
    TDSA_FOR_SIZES(TDSA_CASE)
#undef TDSA_CASE
    default: return LaunchGeom{1, 256, 1, 0};
  }
}

hipError_t launch_window_perm(int log2n, const float* w, float* wp, hipStream_t s) {
  switch (log2n) {
#define TDSA_CASE(k) case k: return perm_size<k>(w, wp, s);
    TDSA_FOR_SIZES(TDSA_CASE)
#undef TDSA_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_spectrum(int log2n, int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  switch (log2n) {
#define TDSA_CASE(k) case k: return launch_size<k>(in_c64, p, g, s);
    TDSA_FOR_SIZES(TDSA_CASE)
#undef TDSA_CASE
    default: return hipErrorInvalidValue;
  }
}
}  // namespace tdsa
