/* harp_hip.h — C ABI of libharp_hip.so: the MI355X (gfx950) kernels of HARP's render-and-compare hot path.
 *
 * The reference (korrawe/harp) has no FFI layer of its own: its hot path reaches native code through PyTorch3D's
 * torch-extension ops and torch operators.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to the reference repo) — this is what a maintainer would bind.
 *
 * Conventions: plain device pointers + sizes; float32 / int32, contiguous row-major; no hidden allocation (outputs and
 * workspaces are caller-provided); every call only enqueues work on `stream` (hipStream_t passed as void*-compatible
 * handle) and returns 0 on success (1 = bad argument, 2+hipError = launch failure); nothing throws across the ABI.
 * "(+=)" marks accumulate-into outputs (caller zero-initialises).
 */
#ifndef HARP_HIP_H
#define HARP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ---- rasteriser -------------------------------------------------------------------------------------------------
 * replaces pytorch3d _C.rasterize_meshes (+ SoftSilhouetteShader / sigmoid_alpha_blend when soft != 0):
 *   MeshRasterizer built at renderer/renderer_helper.py:52-55 (K=50, blur), :76-79 and :444-447 (K=1), invoked at
 *   :344 (light view) and :353 (camera view); SoftSilhouetteShader at :56.
 * ndc (B,V,3) = (x_ndc, y_ndc, z_view) from harp_project_fwd; faces (F,3) shared by all frames.
 * outputs (B,S,S): face_id (frame-local face index of the nearest face, -1 = empty), zbuf (NULL ok; -1 = empty),
 * alpha (soft only).  ws: harp_rasterize_ws_bytes(B,F,S) bytes, 256-B aligned, kept until the backward calls ran. */
size_t harp_rasterize_ws_bytes(int B, int F, int S);
int harp_rasterize_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                       float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, hipStream_t stream);
/* replaces _C.rasterize_meshes_backward (grad_dists path) + sigmoid_alpha_blend backward; g_ndc (B,V,3) (+=) */
int harp_silhouette_bwd(const int32_t* faces, int B, int V, int F, int S, float blur_radius, float sigma, const void* ws,
                        const float* alpha, const float* g_alpha, float* g_ndc, hipStream_t stream);
/* replaces _C.rasterize_meshes_backward (grad_zbuf path) for a K=1 pass: g_z (B,S,S) -> g_ndc (B,V,3) (+=) */
int harp_depth_bwd(const int32_t* face_id, const void* ws, const int32_t* faces, const float* g_z, int B, int V, int F, int S,
                   float* g_ndc, hipStream_t stream);

/* ---- shader --------------------------------------------------------------------------------------------------------
 * replaces SoftPhongShaderShadow.forward + phong_shading_with_shadow + the shadow-map test of
 * MeshRendererShadow.forward (renderer/renderer_helper.py:360-408, 472-523, 565-592), PBRMaterials.apply_normal_map
 * (renderer/pbr_materials.py:58-124), TexturesUV.sample_textures, interpolate_face_attributes, _apply_lighting and
 * softmax_rgb_blend; with zl == NULL it is SoftPhongShaderPBR / phong_shading_PBR (renderer_helper.py:106-190). */
typedef struct harp_shade_args {
  /* geometry of the camera-view K=1 pass */
  const int32_t* face_id;   /* (B,S,S) from harp_rasterize_fwd */
  const void* recs;         /* the workspace of that call */
  const int32_t* faces;     /* (F,3) */
  const int32_t* faces_uvs; /* (F,3) */
  const float* verts_uvs;   /* (VT,2) */
  const float* verts;       /* (B,V,3) world */
  const float* vnormals;    /* (B,V,3) unit vertex normals */
  /* appearance */
  const float* tex;         /* (Ht,Wt,3) albedo */
  const float* nmap;        /* (Ht,Wt,3) per-texel normalised normal map, or NULL */
  const float* light_pos;   /* (B,3) */
  const float* colors;      /* 9 floats on device: ambient rgb, diffuse rgb, specular rgb (light x material) */
  /* self shadow (NULL zl => no shadow term) */
  const float* zl;          /* (B,S,S) light-view depth map (-1 empty) */
  const float* light_R;     /* (B,9) row-major, X_view = X R + T */
  const float* light_T;     /* (B,3) */
  int B, V, F, S, Ht, Wt;
  float focal, ppx, ppy;
  float bg[3];
  float* rgb;               /* out (B,S,S,3) */
  /* backward only */
  const float* g_rgb;       /* (B,S,S,3) */
  float* g_tex;             /* (Ht,Wt,3) (+=) or NULL */
  float* g_nmap;            /* (Ht,Wt,3) (+=) or NULL */
  float* g_verts;           /* (B,V,3) (+=) */
  float* g_vnormals;        /* (B,V,3) (+=) */
  float* g_ndc;             /* (B,V,3) (+=) camera-view NDC vertices */
  float* g_zl;              /* (B,S,S) (+=) or NULL */
  float* g_light_pos;       /* (B,3) (+=) or NULL */
  float* g_colors;          /* 9 (+=) or NULL */
  float* g_light_R;         /* (B,9) (+=) or NULL */
  float* g_light_T;         /* (B,3) (+=) or NULL */
} harp_shade_args;
int harp_shade_fwd(const harp_shade_args* a, hipStream_t stream);
int harp_shade_bwd(const harp_shade_args* a, hipStream_t stream);

/* ---- mesh preparation --------------------------------------------------------------------------------------------
 * replaces the PyTorch3D object churn of utils/visualize.py:prepare_mesh (:45-64): Meshes(...), SubdivideMeshes
 * (optimize_sequence.py:67-89), verts_normals_padded, displacement; and Meshes.verts_normals_packed in the shaders
 * (renderer_helper.py:495).  CSR tables come from harp_amd/topology.py. */
int harp_subdivide_fwd(const float* v0, const int32_t* edges0, int B, int V0, int E0, float scale, float* vs, hipStream_t stream);
int harp_subdivide_bwd(const float* g_vs, const int32_t* sub_off, const int32_t* sub_idx, int B, int V0, int V, float scale,
                       float* g_v0, hipStream_t stream);
int harp_vertex_normals_fwd(const float* v, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_idx, int B, int V,
                            float* n, float* inv_len, const float* disp, float* vd, hipStream_t stream);
int harp_vertex_normals_bwd(const float* v, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_idx, int B, int V,
                            const float* n, const float* inv_len, const float* g_n, float* tmp, float* g_v, hipStream_t stream);
int harp_displace_bwd(const float* g_vd, const float* n, const float* disp, int B, int V, float* g_n, float* g_disp,
                      hipStream_t stream);
/* replaces MeshRasterizer.transform with PerspectiveCameras(in_ndc=False) (call sites utils/visualize.py:272-279, 304-313) */
int harp_project_fwd(const float* v, const float* R, const float* T, int B, int V, float focal, float ppx, float ppy, int S,
                     float* ndc, hipStream_t stream);
int harp_project_bwd(const float* v, const float* R, const float* T, const float* g_ndc, int B, int V, float focal, int S,
                     float* g_v, float* g_R, float* g_T, hipStream_t stream);
/* hand_verts.mean(1) (optimize_sequence.py:476) */
int harp_centroid(const float* v, int B, int V, float* c, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HARP_HIP_H */
