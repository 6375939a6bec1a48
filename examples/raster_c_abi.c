/* The C ABI of libharp_hip.so from plain C: no Python, no torch, no C++ — device pointers and sizes only.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/raster_c_abi.c \
 *       -L harp_amd/csrc -lharp_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/harp_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/raster_c_abi
 *
 * Rasterises two frames of a two-triangle mesh at 128x128 with the hard K = 1 pass (what PyTorch3D's
 * _C.rasterize_meshes(faces_per_pixel=1, blur_radius=0) returns as pix_to_face / zbuf, renderer_helper.py:76-79) and with the
 * soft silhouette (K = 50, SoftSilhouetteShader, renderer_helper.py:44-58), then prints the covered-pixel counts and checks them
 * against the triangle areas.  Exit code 0 = ok.  (tests/test_c_example.py compiles it on the CPU and runs it on the GPU box.) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "harp_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HARP(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s returned %d\n", #x, r_); return 3; } } while (0)

int main(void) {
  enum { B = 2, V = 4, F = 2, S = 128 };
  /* NDC vertices (x, y, z_view), PyTorch3D convention (+x left, +y up); frame 1 is frame 0 shrunk by half */
  float ndc[B][V][3] = {{{0.6f, 0.6f, 2.f}, {-0.6f, 0.6f, 2.f}, {-0.6f, -0.6f, 2.f}, {0.6f, -0.6f, 3.f}},
                        {{0.3f, 0.3f, 2.f}, {-0.3f, 0.3f, 2.f}, {-0.3f, -0.3f, 2.f}, {0.3f, -0.3f, 3.f}}};
  int32_t faces[F][3] = {{0, 1, 2}, {0, 2, 3}};
  float *d_ndc, *d_z, *d_alpha;
  int32_t *d_faces, *d_fid;
  void* d_ws;
  const size_t npix = (size_t)B * S * S, ws_bytes = harp_rasterize_ws_bytes(B, F, S);
  CHECK_HIP(hipMalloc((void**)&d_ndc, sizeof ndc));
  CHECK_HIP(hipMalloc((void**)&d_faces, sizeof faces));
  CHECK_HIP(hipMalloc((void**)&d_fid, npix * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_z, npix * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_alpha, npix * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
  CHECK_HIP(hipMemcpy(d_ndc, ndc, sizeof ndc, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_faces, faces, sizeof faces, hipMemcpyHostToDevice));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  int32_t* fid = (int32_t*)malloc(npix * sizeof(int32_t));
  float* alpha = (float*)malloc(npix * sizeof(float));
  float* z = (float*)malloc(npix * sizeof(float));
  /* hard pass: soft = 0 */
  CHECK_HARP(harp_rasterize_fwd(d_ndc, d_faces, B, V, F, S, 0, 0.f, 1.f, d_ws, d_fid, d_z, NULL, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(fid, d_fid, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(z, d_z, npix * sizeof(float), hipMemcpyDeviceToHost));
  long covered[B] = {0, 0};
  float zmin = 1e30f, zmax = -1e30f;
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < (size_t)S * S; ++i)
      if (fid[(size_t)b * S * S + i] >= 0) {
        ++covered[b];
        const float zz = z[(size_t)b * S * S + i];
        if (zz < zmin) zmin = zz;
        if (zz > zmax) zmax = zz;
      }
  /* the square spans 1.2 (0.6) NDC units of 2 -> (0.6 S)^2 and (0.3 S)^2 pixels, up to the edge pixels */
  const double want0 = 0.6 * S * 0.6 * S, want1 = 0.3 * S * 0.3 * S;
  printf("hard pass : covered %ld / %ld pixels (expected ~%.0f / ~%.0f), depth range [%.3f, %.3f]\n", covered[0], covered[1], want0, want1, zmin, zmax);
  int ok = fabs(covered[0] - want0) < 4 * 0.6 * S && fabs(covered[1] - want1) < 4 * 0.3 * S && zmin >= 2.f - 1e-4f && zmax <= 3.f + 1e-4f;

  /* soft silhouette: soft = 1, blur radius and sigma as in renderer_helper.get_renderers(silh_sigma = 1e-7) */
  const float sigma = 1e-7f, blur = logf(1.f / 1e-4f - 1.f) * sigma;
  CHECK_HARP(harp_rasterize_fwd(d_ndc, d_faces, B, V, F, S, 1, blur, sigma, d_ws, d_fid, NULL, d_alpha, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(alpha, d_alpha, npix * sizeof(float), hipMemcpyDeviceToHost));
  double asum[B] = {0, 0};
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < (size_t)S * S; ++i) asum[b] += alpha[(size_t)b * S * S + i];
  printf("soft pass : sum(alpha) %.1f / %.1f\n", asum[0], asum[1]);
  ok = ok && fabs(asum[0] - want0) < 4 * 0.6 * S && fabs(asum[1] - want1) < 4 * 0.3 * S;
  /* argument checking: a NULL output is an error code, not a crash */
  ok = ok && harp_rasterize_fwd(d_ndc, d_faces, B, V, F, S, 0, 0.f, 1.f, d_ws, NULL, d_z, NULL, stream) == 1;
  printf(ok ? "ok\n" : "MISMATCH\n");
  hipStreamDestroy(stream);
  hipFree(d_ndc); hipFree(d_faces); hipFree(d_fid); hipFree(d_z); hipFree(d_alpha); hipFree(d_ws);
  free(fid); free(alpha); free(z);
  return ok ? 0 : 1;
}
