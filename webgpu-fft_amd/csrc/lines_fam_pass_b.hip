// device code + launch stubs of the FAM_PASS_B line-FFT kernels (see dispatch.hpp / line_kernels.def)
#include "hip_launcher.hpp"
namespace mi355 {
template bool launch_lines_family<FAM_PASS_B, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);
}
