// Per-frame fused mesh chain for gfx950.  In a captured step every kernel node costs ~4.5 us of dispatch latency however small
// it is, and the geometry between LBS and the rasterisers was 9 such nodes forward (metre scaling, SubdivideMeshes, vertex
// normals, displacement, second normals, centroid, light camera, two projections) and 13 backward — ~35 % of a 1.5-ms step
// (profiles/r01_e_timeline_one_step.txt).  One frame's mesh is 3-4 k vertices = 37-49 KB: it fits the LDS of one CU, so each
// chain becomes ONE launch, one 1024-thread workgroup per frame, stages separated by workgroup barriers instead of kernel
// boundaries.  Same arithmetic as the stand-alone kernels in mesh.hip / glue.hip (which stay as the C-ABI building blocks):
//   forward  : utils/visualize.py:45-64 (prepare_mesh), renderer_helper.py:454-468 (process_info_for_shadow),
//              MeshRasterizer.transform for both views, hand_verts.mean(1) (optimize_sequence.py:476)
//   backward : the autograd of the above.
#include "chain_body.h"

namespace {

using namespace cb;

__global__ void __launch_bounds__(kChainThreads) mesh_chain_fwd_kernel(const harp_mesh_chain A) {
  extern __shared__ float s_p[];
  const int b = blockIdx.x;
  mesh_chain_fwd_body(A, s_p, b, false, A.cam_R + 9 * b, A.cam_T + 3 * b, A.shadow ? A.light_pos + 3 * b : nullptr);
}

__global__ void __launch_bounds__(kChainThreads) mesh_chain_bwd_kernel(const harp_mesh_chain A) {
  extern __shared__ float s_mem[];
  mesh_chain_bwd_body(A, s_mem, blockIdx.x);
}

}  // namespace

extern "C" {

static bool chain_ok(const harp_mesh_chain* a) {
  return a && a->edges0 && a->vf_off && a->vf_tri && a->disp && a->B > 0 && a->V0 > 0 && a->E0 >= 0 &&
         a->V0 + a->E0 <= kChainThreads * kMaxPerThread && a->NJ * 3 <= kChainThreads;
}

int harp_mesh_chain_max_vertices(void) { return kChainThreads * kMaxPerThread; }

int harp_mesh_chain_fwd(const harp_mesh_chain* a, hipStream_t stream) {
  if (!chain_ok(a) || !a->verts_mm || !a->joints_mm || !a->cam_R || !a->cam_T || !a->joints_m || !a->vs || !a->n1 || !a->il1 || !a->vd ||
      !a->n2 || !a->il2 || !a->ndc_c || (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->ndc_l)))
    return HARP_ERR_ARG;
  const size_t lds = (size_t)(a->V0 + a->E0) * 3 * sizeof(float);
  hipLaunchKernelGGL(mesh_chain_fwd_kernel, dim3(a->B), dim3(kChainThreads), lds, stream, *a);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_mesh_chain_bwd(const harp_mesh_chain* a, hipStream_t stream) {
  if (!chain_ok(a) || !a->sub_off || !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R || !a->cam_T || !a->g_vd ||
      !a->g_ndc_c || !a->g_joints_m || !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp ||
      (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  const size_t lds = (size_t)(a->V0 + a->E0) * 9 * sizeof(float);
  // dynamic LDS above 64 KB has to be requested (3 buffers of V*12 B: 111 KB hand, 147 KB arm); per-device attribute, set on every call
  if (hipFuncSetAttribute((const void*)mesh_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(mesh_chain_bwd_kernel, dim3(a->B), dim3(kChainThreads), lds, stream, *a);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
