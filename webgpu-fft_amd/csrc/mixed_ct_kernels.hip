// device code + launch stubs of the mixed-radix line kernels with compile-time radix plans (kern_mixed_ct.hpp)
#define MI355_MIXEDCT_DEFINE_INSTANCES
#include "hip_launcher.hpp"
namespace mi355 {
template bool launch_mixedct<HipLauncher>(int, const MixedArgs&, unsigned, HipLauncher&);
template bool launch_line_reg<HipLauncher>(int, const MixedArgs&, unsigned, HipLauncher&);
}
