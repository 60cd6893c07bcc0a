// device code + launch stub of the XCD-resident N = 1024 x 1024 kernel (kern_xcd_res.hpp), forward / inverse / skeletons
#define MI355_XCD_RES_DEFINE_INSTANCES
#include "hip_launcher.hpp"
namespace mi355 {
template bool launch_xcd_res<HipLauncher>(int, const XcdFusedArgs&, unsigned, HipLauncher&);
}
