// device code + launch stubs of the XCD-fused four-step kernels (kern_xcd.hpp, dispatch.hpp launch_xcd_fused)
#include "hip_launcher.hpp"
namespace mi355 {
template bool launch_xcd_fused<HipLauncher>(int, const XcdFusedArgs&, unsigned, HipLauncher&);
}
