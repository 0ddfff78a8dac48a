// Compiles integration/DmsaOptimizerHip.h against the REAL Eigen / PCL headers and the reference's own classes (no GPU, no linking): every
// member the binding touches exists with the type it assumes.  Built by oracle/ref_harness/CMakeLists.txt on a machine that has them.
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "PointStampId.h"
#include "DmsaOptimizerHip.h"

void dmsa_binding_check(ContinuousTrajectory& traj, MapManagement& map) {
    DmsaOptimSettings s;
    DmsaOptimizerHip<PointStampId> window;
    window.optimizeSet(traj, s);
    DmsaOptimizerHip<pcl::PointNormal> keyframes;
    keyframes.optimizeSet(map, s);
}
