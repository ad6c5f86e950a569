// temporary stubs until the Brillig kernels land
#include "kernels.hpp"
namespace acvm {
void launch_brillig_level(hipStream_t, uint4 *, uint64_t, uint32_t, const DeviceProgram &, const uint32_t *, const uint32_t *, uint32_t, uint32_t *, uint32_t *) {}
void launch_exact_brillig(hipStream_t, uint4 *, uint64_t, const DeviceProgram &, const ExactLanes &, uint32_t, uint32_t *) {}
}
