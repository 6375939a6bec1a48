/* harp_hip.h — C ABI of libharp_hip.so: the MI355X (gfx950) kernels of HARP's render-and-compare hot path.
 *
 * The reference (korrawe/harp) has no FFI layer of its own: its hot path reaches native code through PyTorch3D's
 * torch-extension ops and torch operators.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to the reference repo) — this is what a maintainer would bind.
 *
 * Conventions: plain device pointers + sizes; float32 / int32, contiguous row-major; no hidden allocation (outputs and
 * workspaces are caller-provided); every call only enqueues work on `stream` (hipStream_t passed as void*-compatible
 * handle) and returns 0 on success (1 = bad argument, 2+hipError = launch failure, 1000 = RCCL not found, 1001+ncclResult = RCCL
 * error); nothing throws across the ABI.
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
/* The set-up part of the rasteriser calls below (face records with the blur-dilated bbox, per-super-tile face lists, launch order:
 * PyTorch3D's coarse rasterisation, renderer/renderer_helper.py:52-58, :344, :353) for TWO views of the same meshes in three launches
 * instead of six.  A rasteriser call on a workspace prepared here passes bit 2 (value 4) in its `soft` / `sparse` argument and then only
 * launches its tile pass.  blur_radius_x = the blur_radius of that view's pass (0 for a hard K = 1 pass). */
int harp_raster_setup_pair(const float* ndc_a, float blur_radius_a, void* ws_a, const float* ndc_b, float blur_radius_b, void* ws_b,
                           const int32_t* faces, int B, int V, int F, int S, hipStream_t stream);
/* K = 1 depth pass (the light view, renderer_helper.py:344) into a depth map zbuf that the caller KEEPS between calls.  st_state:
 * B * ceil(S/64)^2 ints, zero before the first call and owned by the library afterwards — which 64x64 super-tiles of zbuf hold -1
 * everywhere.  With sparse != 0 (face ids of super-tiles without a face are not written, as with soft bit 1 of harp_rasterize_fwd) a
 * super-tile that is empty again is not filled again: in a fitting step that is 3/4 of the map.  Nobody else may write zbuf between calls. */
int harp_rasterize_fwd_keep(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int sparse, void* ws, int32_t* face_id,
                            float* zbuf, int32_t* st_state, hipStream_t stream);
/* the same with torch.nn.L1Loss(y_sil_true, y_sil_pred) (optimize_sequence.py:519) fused into the raster epilogue: l1_target (T,S,S)
 * indexed by l1_fid (B,), *l1_loss (+=) the mean, l1_grad (B,S,S) = l1_w[0] * d loss / d alpha.  l1_target == NULL: plain rasterisation. */
/* soft bit 0: soft silhouette on.  soft bit 1 (value 2): SPARSE outputs — face_id / alpha / l1_grad are left unwritten in 64x64
 * super-tiles that hold no face (their contribution to the loss is still accumulated; zbuf is always written everywhere).  For callers whose consumers skip those super-tiles too
 * (harp_shade_*, harp_silhouette_bwd do): 3/4 of a hand image is such background. */
/* l1_bg_sums (optional, used with sparse outputs): (T, nsx*nsx) per target frame and 64x64 super-tile, the term's sum over the
 * super-tile when nothing is rendered there (sum of y_sil) — targets are static during a fit, so the background part of the loss is
 * a table look-up instead of a pass over 3/4 of the target image every step. */
/* face_id == NULL (with the soft silhouette on and zbuf == NULL): silhouette only — the nearest face per pixel is neither formed nor
 * written (the geometry-only stage of a fit reads alpha alone). */
int harp_rasterize_l1_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                          float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, const float* l1_target,
                          const int32_t* l1_fid, const float* l1_w, float* l1_loss, float* l1_grad, const float* l1_bg_sums,
                          hipStream_t stream);
/* harp_rasterize_l1_fwd (soft & 1, y_sil != NULL) with the silhouette backward fused into the SAME launch (renderer_helper.py:44-58 forward,
 * rasterize_meshes_backward's dists path + sigmoid_alpha_blend's backward): g_ndc (B,V,3) += d (w * L1(alpha, y_sil)) / d ndc, x and y
 * components — what harp_silhouette_bwd(alpha, g_alpha) of the same workspace would add.  A tile's rim pixels walk the tile's faces again
 * while these are still staged in LDS; no second launch re-reads alpha / g_alpha and re-stages the lists.  alpha and g_alpha are still written. */
int harp_rasterize_l1_fwd_bwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius, float sigma,
                              void* ws, int32_t* face_id, float* alpha, const float* y_sil, const int32_t* fid, const float* w, float* loss,
                              float* g_alpha, const float* bg_sums, float* g_ndc, hipStream_t stream);
/* replaces _C.rasterize_meshes_backward (grad_dists path) + sigmoid_alpha_blend backward; g_ndc (B,V,3) (+=) */
int harp_silhouette_bwd(const int32_t* faces, int B, int V, int F, int S, float blur_radius, float sigma, const void* ws,
                        const float* alpha, const float* g_alpha, float* g_ndc, hipStream_t stream);
/* replaces _C.rasterize_meshes_backward (grad_zbuf path) for a K=1 pass: g_z (B,S,S) -> g_ndc (B,V,3) (+=) */
int harp_depth_bwd(const int32_t* face_id, const void* ws, const int32_t* faces, const float* g_z, int B, int V, int F, int S,
                   float* g_ndc, hipStream_t stream);
/* the same for a caller that keeps ONE gradient image g_z across steps (the fitting loop: harp_shade_bwd scatters the shadow test's tap
 * gradients of renderer_helper.py:385-408 into it): every non-zero entry read is also cleared.  The shader only ever writes at light-view
 * pixels that hold a face (an empty texel's depth -1 gives a shadow-test sigmoid of exactly 0), all of which this pass visits, so an
 * image that was all-zero before the step's harp_shade_bwd is all-zero again after this call and needs no per-step clear. */
int harp_depth_bwd_consume(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S,
                           float* g_ndc, hipStream_t stream);
/* harp_depth_bwd_consume + harp_normalize3_bwd(nmap, g_nmap_n, n_texels, g_nmap) as ONE launch: the two small passes between the shader
 * backward and the per-frame backward tail of a fitting step (neither reads what the other writes) */
int harp_depth_nmap_bwd(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S, float* g_ndc,
                        const float* nmap, const float* g_nmap_n, int n_texels, float* g_nmap, unsigned char* g_z_tiles, hipStream_t stream);
/* harp_depth_bwd_consume restricted to the tiles flagged in g_z_tiles (harp_shade_args.g_zl_tiles of the backward that filled g_z; NULL:
 * every tile); the flags of the tiles visited are cleared */
/* g_vert9 (n_verts, 9) of harp_shade_args -> g_verts / g_vnormals / g_ndc (n_verts,3 each) +=, g_vert9 all-zero again */
int harp_vert9_unpack(float* g_vert9, int n_verts, float* g_verts, float* g_vnormals, float* g_ndc, hipStream_t stream);
/* harp_depth_bwd_tiles (g_z consumed; g_z_tiles optional) with up to two riders in the SAME launch (extra workgroups behind the tile workgroups:
 * both only need the shader backward's output): nmap != NULL: harp_normalize3_bwd(nmap, g_nmap_n, n_texels, g_nmap); g_vert9 != NULL:
 * harp_vert9_unpack(g_vert9, B * V, g_verts, g_vnormals, g_ndc_cam). */
int harp_depth_bwd_riders(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S, float* g_ndc,
                          unsigned char* g_z_tiles, const float* nmap, const float* g_nmap_n, int n_texels, float* g_nmap, float* g_vert9,
                          float* g_verts, float* g_vnormals, float* g_ndc_cam, hipStream_t stream);
int harp_depth_bwd_tiles(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S,
                         float* g_ndc, unsigned char* g_z_tiles, hipStream_t stream);

/* ---- fragment-level rasterisation (the PyTorch3D op pair itself; NOT on the fitting loop's path) -----------------------------
 * replaces _C.rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
 *   blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct=True, clip_barycentric_coords=(blur_radius>0),
 *   cull_backfaces=False) -> (pix_to_face, zbuf, bary, dists) and _C.rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf,
 *   grad_bary, grad_dists, perspective_correct, clip_barycentric_coords) -> grad_face_verts, as reached through MeshRasterizer at
 *   renderer/renderer_helper.py:52-55 (K=50, blur), :76-79 (K=1), :88-101 (K=10 normal renderer), :444-447.
 * For callers that keep PyTorch3D-style shader classes.  K = faces_per_pixel is a CAP (1 <= K <= 64): the K nearest candidates are
 * kept in ascending depth, ties keep the lower face index.  Outputs (B,S,S,K): pix_to_face int32 frame-local (-1 = empty slot; the
 * PyTorch3D "packed" index is b*F + f), zbuf, bary (B,S,S,K,3), dists (signed squared NDC distance to the nearest edge); empty slots
 * hold -1 everywhere.  ws: harp_rasterize_ws_bytes(B,F,S).  Backward: any of g_zbuf / g_bary / g_dists may be NULL; g_ndc (+=). */
int harp_rasterize_fragments_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, float blur_radius, int K, void* ws,
                                 int32_t* pix_to_face, float* zbuf, float* bary, float* dists, hipStream_t stream);
int harp_rasterize_fragments_bwd(const float* ndc, const int32_t* faces, const int32_t* pix_to_face, const float* g_zbuf,
                                 const float* g_bary, const float* g_dists, int B, int V, int F, int S, float blur_radius, int K,
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
  float* rgb;               /* out (B,S,S,3); may be NULL when l1_target is set: the loss and its gradient are produced without
                             * materialising the image (4 MB per 512x512 frame that nothing in an optimisation step reads back) */
  /* backward only */
  const float* g_rgb;       /* (B,S,S,3) */
  float* g_tex;             /* (Ht,Wt,3) (+=) or NULL */
  float* g_nmap;            /* (Ht,Wt,3) (+=) or NULL */
  float* g_verts;           /* (B,V,3) (+=) */
  float* g_vnormals;        /* (B,V,3) (+=) */
  float* g_ndc;             /* (B,V,3) (+=) camera-view NDC vertices.  harp_shade_bwd: all three NULL = no geometry gradients (the
                             * appearance-only stage, optimize_sequence.py:264-310: opt_app holds texture, normal map, light, ambient ratio) */
  float* g_zl;              /* (B,S,S) (+=) or NULL; see g_zl_tiles at the end of the struct */
  float* g_light_pos;       /* (B,3) (+=) or NULL */
  float* g_colors;          /* 9 (+=) or NULL */
  float* g_light_R;         /* (B,9) (+=) or NULL */
  float* g_light_T;         /* (B,3) (+=) or NULL */
  int debug_skip;           /* 0 in production (wave-autonomous backward kernel, csrc/shade_bwd.hip).  Non-zero selects the first,
                             * barrier-synchronised backward kernel: 64 = that kernel unmodified (A/B timing, same results); bits 0-5 are
                             * its ablation switches (tools/dev/gpu_shade_r3.py): 1/2 no texel adds, 4 no shadow-tap gradient, 8 no vertex
                             * adds, 16 no texel flush, 32 no vertex flush — results are then WRONG by design.  Bits 8-13 are the same kind of
                             * switches for the production kernel: 256 no texel phase, 512 no shadow window, 1024 no vertex phase, 2048 no table
                             * flushes, 8192 no shading at all (fixed cost of the launch) */
  /* forward only, optional: fused photometric L1 (optimize_sequence.py:543): *l1_loss (+=) mean |y_pred*m - y_true[fid]*m|,
   * l1_grad (B,S,S,3) = l1_w[0] * d loss / d y_pred, WRITTEN ONLY AT COVERED PIXELS (face_id >= 0: the only ones harp_shade_bwd reads).
   * l1_target == NULL disables it. */
  const float* l1_target;   /* (T,S,S,3) */
  const float* l1_mask;     /* (T,S,S) or NULL */
  const int32_t* l1_fid;    /* (B,) */
  const float* l1_w;
  float* l1_loss;
  float* l1_grad;
  float l1_inv;             /* filled in by harp_shade_fwd */
  /* optional (fwd and bwd): the SAME tex / nmap interleaved by harp_pack_texels; halves the cache lines of the bilinear footprint */
  const void* texnm;
  /* optional, used when rgb == NULL: (T, nsx*nsx) per target frame and 64x64 super-tile, sum over its pixels and channels of
   * |bg_c - y_c| * mask — the photometric term of a super-tile that holds no face (static targets: a table instead of reading mask
   * and target of 3/4 of the image every step) */
  const float* l1_bg_sums;
  /* optional (backward): (B, ceil(S/16), ceil(S/16)) bytes; the byte of every 16x16 light-view tile in which the backward adds to g_zl
   * is set to 1.  harp_depth_bwd_tiles / harp_depth_nmap_bwd read them to leave out the tiles without a gradient (about half of the
   * tiles they would otherwise read) and clear the ones they consume. */
  unsigned char* g_zl_tiles;
  /* optional (harp_shade_bwd, production kernel): texel-gradient RECORDS.  With trec != NULL the pass does not scatter the bilinear
   * footprints of g_tex / g_nmap (the backward of TexturesUV.sample_textures, renderer/pbr_materials.py:82-124) itself: every shaded
   * pixel appends ONE 36-byte record (texel x0 | y0 << 16, the two bilinear fractions, g_albedo[3], g_nmap[3]) to the list of the
   * 32x32-texel UV tile its top-left texel lies in, and harp_texel_reduce (below) adds the lists up tile by tile in LDS into the double
   * maps trec_acc_tex / trec_acc_nmap, which harp_texel_finish turns into g_tex / g_nmap.  harp_shade_bwd itself leaves g_tex / g_nmap
   * untouched (they still say WHICH maps take a gradient: NULL = frozen); a record that does not fit its list is added to the double maps
   * with memory atomics straight away.
   *   trec      (harp_texel_bins(Ht, Wt), 9, trec_cap) floats, plane k of bin b at ((b * 9 + k) * trec_cap); 16-byte aligned, trec_cap a
   *             multiple of 4
   *   trec_cnt  harp_texel_bins(Ht, Wt) * 16 + 16 int32: the record count of bin b at [16 b] (one counter per 64-byte line), ALL ZERO on
   *             entry; harp_texel_reduce hands them back zeroed */
  float* trec;
  int32_t* trec_cnt;
  int trec_cap;
  double* trec_acc_tex;    /* (Ht,Wt,3) doubles, the accumulators harp_texel_reduce adds into (required with trec unless the map is frozen) */
  double* trec_acc_nmap;
  /* optional (harp_shade_bwd, production kernel): (B,V,9) (+=).  The three vertex gradients go HERE instead of into g_verts / g_vnormals /
   * g_ndc, interleaved per vertex [g_verts(3) | g_vnormals(3) | g_ndc(3)]: a wave's table flush is then one 36-byte run per vertex instead of
   * three 12-byte runs in three arrays — a third of the memory-atomic lines.  g_verts / g_vnormals / g_ndc must still be non-NULL (they say
   * that geometry gradients are wanted) and stay untouched; harp_vert9_unpack or harp_depth_bwd_riders adds the buffer into them and clears it. */
  float* g_vert9;
} harp_shade_args;
/* number of 32x32-texel UV tiles (record bins) of an (Ht, Wt) map */
int harp_texel_bins(int Ht, int Wt);
/* second half of the texel gradient (see harp_shade_args.trec): acc_tex / acc_nmap (Ht,Wt,3) DOUBLE += the bilinear footprints of the
 * records the harp_shade_bwd call(s) since the last reduce appended.  One workgroup per chunk of 2048 records of a bin: a (33 x 33 texel) x
 * 6 channel 64-bit fixed-point accumulator in LDS (the extra row / column takes the footprints of the tile's last row / column; sums
 * exact and independent of the order of the records), added to the double maps with row-contiguous memory atomics — the gradient a texel
 * ends up with is float(exact sum) however the frames were batched.  Either map pointer may be NULL (frozen map).  The counters are
 * all-zero again when the call has run.  expected_records: a hint for the launch shape only (<= 2 M or <= 0: one workgroup per CU, which leaves
 * LDS for the kernels that run beside it; more: two) — e.g. a sixth of B * S * S. */
int harp_texel_reduce(const float* trec, int32_t* trec_cnt, int trec_cap, int Ht, int Wt, double* acc_tex, double* acc_nmap, int expected_records,
                      hipStream_t stream);
/* third and last part: g_tex (n_texels,3) += float(acc_tex), g_nmap += float(acc_nmap) — with nmap_raw != NULL through the chain rule of
 * F.normalize(nmap_raw, dim=-1) (utils/visualize.py:99; harp_normalize3_bwd's arithmetic: acc_nmap is then the gradient of the NORMALISED
 * map, g_nmap that of the raw one) — and both accumulators are all-zero again.  A NULL accumulator skips that map. */
int harp_texel_finish(double* acc_tex, float* g_tex, double* acc_nmap, float* g_nmap, const float* nmap_raw, int n_texels, hipStream_t stream);
/* interleaves albedo (Ht*Wt,3) and the normalised normal map (Ht*Wt,3) into out (Ht*Wt,8): [r g b nx | ny nz 0 0], 16-B aligned */
int harp_pack_texels(const float* tex, const float* nmap, int n_texels, float* out, hipStream_t stream);
/* harp_normalize3_fwd(nmap_raw) -> nmap_n and harp_pack_texels(tex, nmap_n) -> packed (may be NULL) in ONE launch */
int harp_normalize3_pack(const float* tex, const float* nmap_raw, int n_texels, float* nmap_n, float* packed, hipStream_t stream);
int harp_shade_fwd(const harp_shade_args* a, hipStream_t stream);
/* backward of the shading pass.  g_rgb != NULL: plain backward of an upstream gradient image.  g_rgb == NULL (FUSED-LOSS mode, needs
 * l1_target / l1_fid / l1_w / l1_loss / l1_bg_sums): no harp_shade_fwd call is needed at all — the pass recomputes the colour anyway,
 * forms torch.nn.L1Loss(y_true * m, y_pred * m) (optimize_sequence.py:543) and its gradient itself and accumulates the loss value;
 * with a->rgb != NULL it also writes that colour, i.e. the whole rendered image y_pred (background included), as harp_shade_fwd would. */
int harp_shade_bwd(const harp_shade_args* a, hipStream_t stream);
/* harp_shade_bwd(a) and harp_silhouette_bwd(a->faces, ..., ws = a->recs, alpha, g_alpha, g_ndc = a->g_ndc) of the SAME camera-view
 * rasterisation as ONE launch whose workgroups alternate between the two kinds of tile: as separate kernels on two streams they cannot
 * share a CU (registers / LDS), in one grid the rasteriser's waves issue while the shader's wait on memory.  Same results.  (a->trec is
 * ignored: this launch hosts the table form of the shader tile, the texel gradients go straight into g_tex / g_nmap.) */
int harp_shade_sil_bwd(const harp_shade_args* a, float blur_radius, float sigma, const float* alpha, const float* g_alpha,
                       hipStream_t stream);

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

/* ---- MANO linear-blend skinning ----------------------------------------------------------------------------------
 * replaces ManoLayer.forward (manopth/manolayer.py:108-296, rodrigues_layer.py:15-54, tensutils.py:6-42) as built by
 * utils/hand_model_utils.py:74 (use_pca=False, flat_hand_mean=False, axis-angle root, right hand); call site
 * utils/visualize.py:42-44.  Model arrays are device pointers prepared once by harp_amd/manopth/manolayer.py. */
typedef struct harp_mano_model {
  const float* v_template;   /* (778,3) */
  const float* shapedirs_T;  /* (10, 778*3)  th_shapedirs transposed */
  const float* posedirs_T;   /* (135, 778*3) th_posedirs transposed */
  const float* posedirs;     /* (778*3, 135) */
  const float* J_template;   /* (16,3)  = J_regressor @ v_template */
  const float* J_dirs;       /* (16*3,10) = J_regressor @ shapedirs */
  const float* weights;      /* (778,16) */
  const float* hands_mean;   /* (45,) */
} harp_mano_model;
size_t harp_lbs_mano_ws_floats(int B);
/* pose (B,48) = [root axis-angle, 45 hand pose], betas (B,10), trans (B,3) -> verts (B,778,3) mm, joints (B,21,3) mm */
int harp_lbs_mano_fwd(const harp_mano_model* m, const float* pose, const float* betas, const float* trans, int B, float* ws,
                      float* verts, float* joints, hipStream_t stream);
/* ws must be the workspace of the forward call; g_verts is modified in place (tip-joint gradients folded in) */
int harp_lbs_mano_bwd(const harp_mano_model* m, const float* pose, const float* betas, const float* trans, int B, float* ws,
                      float* g_verts, const float* g_joints, float* g_pose, float* g_betas, float* g_trans, hipStream_t stream);

/* ---- kinematic-tree LBS (SMPL-X right arm) ------------------------------------------------------------------------
 * replaces SMPLXARM.forward(..., return_type='mano_w_arm') (hand_models_harp/body_models.py:2163-2390: smplx.lbs :2335, recentre on
 * the right wrist :2342-2343, transl :2378-2380, arm slice :2383-2390); call site utils/visualize.py:37-40.  The model is sliced
 * to the arm vertices on the host (harp_amd/hand_models_harp/body_models.py), the joint regressor folded into J_template/J_dirs. */
typedef struct harp_tree_model {
  int NV, NJ, NB;              /* vertices kept, joints (<= 64), shape coefficients (<= 32) */
  const float* v_template;     /* (NV,3) */
  const float* shapedirs_T;    /* (NB, NV*3) */
  const float* posedirs_T;     /* ((NJ-1)*9, NV*3)  (smplx layout) */
  const float* posedirs;       /* (NV*3, (NJ-1)*9) */
  const float* J_template;     /* (NJ,3) */
  const float* J_dirs;         /* (NJ*3, NB) */
  const float* weights;        /* (NV,NJ) */
  const float* pose_mean;      /* (NJ*3) */
  const int32_t* parents;      /* (NJ) parents[0] = -1, parents[j] < j */
  const int32_t* pose_src;     /* (NJ) row of in_pose driving joint j, or -1 (axis-angle = pose_mean only) */
  int n_pose_in;               /* rows of in_pose: (B, n_pose_in, 3) = [global_orient, right_wrist_pose, right_hand_pose(15)] */
  int center_joint;            /* recentre on this chain joint (21 = right wrist) or -1 */
  int n_joints_out;            /* 22 */
  const int32_t* joint_src;    /* (n_joints_out): >= 0 chain joint id, < 0: vertex joint -(vertex id)-1 */
} harp_tree_model;
size_t harp_lbs_tree_ws_floats(const harp_tree_model* m, int B);
/* verts (B,NV,3) mm, joints (B,n_joints_out,3) mm */
int harp_lbs_tree_fwd(const harp_tree_model* m, const float* in_pose, const float* betas, const float* transl, int B, float* ws,
                      float* verts, float* joints, hipStream_t stream);
/* ws: the workspace harp_lbs_tree_fwd filled for the same inputs AND THE SAME B (the layout of ws is a function of B).  The forward call
 * clears the accumulators this call adds to, and this call leaves them cleared again: it may be repeated on one forward pass, and a
 * workspace may be reused for another batch size by running harp_lbs_tree_fwd at that size first.  Calling it on a workspace whose last
 * forward ran at another B, or that the caller filled itself, accumulates into whatever those bytes hold. */
int harp_lbs_tree_bwd(const harp_tree_model* m, const float* in_pose, const float* betas, const float* transl, int B, float* ws,
                      float* g_verts, const float* g_joints, float* g_in_pose, float* g_betas, float* g_transl, hipStream_t stream);

/* ---- fused per-frame mesh chain -----------------------------------------------------------------------------------
 * One launch (one 1024-thread workgroup per frame, the frame's mesh staged in LDS) for everything utils/visualize.py:45-64
 * (prepare_mesh: /1000, SubdivideMeshes, verts_normals, displacement), Meshes.verts_normals_packed of the displaced mesh
 * (renderer_helper.py:495), hand_verts.mean(1) + process_info_for_shadow (optimize_sequence.py:476, renderer_helper.py:454-468) and
 * MeshRasterizer.transform of both views do between the hand layer and the rasterisers — and one launch for their autograd.
 * Same arithmetic as harp_subdivide_* / harp_vertex_normals_* / harp_displace_bwd / harp_centroid / harp_light_setup_* /
 * harp_project_*; exists because every kernel node of a captured step costs ~4.5 us of dispatch latency.  V0 + E0 <=
 * harp_mesh_chain_max_vertices().  All pointers device memory; "(+=)" outputs accumulate. */
typedef struct harp_mesh_chain {
  const int32_t* edges0;    /* (E0,2) base-mesh edges (midpoint i = V0 + i) */
  const int32_t* vf_off;    /* (V+1) vertex -> incident (face, corner) CSR offsets */
  const int32_t* vf_tri;    /* (nnz,4) per CSR entry: the incident face's vertex ids i0, i1, i2 and this vertex's corner (16-B aligned) */
  const int32_t* sub_off;   /* (V0+1) base vertex -> midpoint CSR (backward) */
  const int32_t* sub_idx;
  const float* disp;        /* (V,) verts_disps */
  int B, V0, E0, NJ, S;
  float focal;
  int shadow;               /* light-view outputs / gradients wanted */
  int has_normal_grad;      /* backward: g_n2 is given (appearance stage) */
  int light_only;           /* backward: ONLY the light-view part — projection backward of g_ndc_l, light camera backward -> g_light_pos (+=);
                             * everything else (camera view, both normal passes, displacement, subdivision, g_v0 / g_joints_mm / g_cam_T /
                             * g_disp) is left out and not written.  The appearance-only stage of a fit: its optimiser holds no geometry
                             * (optimize_sequence.py:264-310), the light position is the one parameter of it this chain feeds. */
  /* forward in */
  const float* verts_mm;    /* (B,V0,3) hand-layer vertices, millimetres */
  const float* joints_mm;   /* (B,NJ,3) */
  const float* cam_R;       /* (B,9) */
  const float* cam_T;       /* (B,3) */
  const float* light_pos;   /* (B,3) */
  /* forward out (backward in) */
  float* joints_m;          /* (B,NJ,3) */
  float* vs;                /* (B,V,3) subdivided, metres */
  float* n1;                /* (B,V,3) its unit vertex normals */
  float* il1;               /* (B,V) 1/|N| (0 where clamped) */
  float* vd;                /* (B,V,3) displaced */
  float* n2;                /* (B,V,3) */
  float* il2;               /* (B,V) */
  float* ndc_c;             /* (B,V,3) camera view */
  float* centroid;          /* (B,3)   shadow only */
  float* light_R;           /* (B,9)   shadow only */
  float* light_T;           /* (B,3)   shadow only */
  float* ndc_l;             /* (B,V,3) shadow only */
  /* backward in */
  const float* g_ndc_c;     /* (B,V,3) */
  const float* g_ndc_l;     /* (B,V,3) shadow only */
  const float* g_n2;        /* (B,V,3) if has_normal_grad */
  const float* g_joints_m;  /* (B,NJ,3) */
  const float* g_vd;        /* (B,V,3) gradient w.r.t. the displaced vertices gathered so far (mesh regularisers, shader) */
  const float* g_light_R;   /* (B,9)  the shader's share, shadow only */
  const float* g_light_T;   /* (B,3) */
  /* backward out */
  float* g_v0;              /* (B,V0,3) w.r.t. verts_mm */
  float* g_joints_mm;       /* (B,NJ,3) */
  float* g_light_pos;       /* (B,3) (+=) shadow only */
  float* g_cam_T;           /* (B,3) (+=) */
  float* g_disp;            /* (V,)  (+=, atomics over frames) */
} harp_mesh_chain;
int harp_mesh_chain_max_vertices(void);
int harp_mesh_chain_fwd(const harp_mesh_chain* a, hipStream_t stream);
int harp_mesh_chain_bwd(const harp_mesh_chain* a, hipStream_t stream);
/* The same forward chain on FOUR workgroups per frame (csrc/chain_wide.hip): each owns a contiguous quarter of the vertices (one per thread),
 * stages the whole frame's positions in LDS, and the two points where a part needs the others' results (displaced neighbours, centroid) are
 * kernel boundaries — two launches of 4 B workgroups instead of one of B.  Same outputs as harp_mesh_chain_fwd (the centroid is the sum of
 * four partial sums).  clear_grads != 0: a->g_vd / a->g_joints_m are zeroed (harp_step_frame.clear_mesh_grads).
 * part_ws: harp_mesh_chain_wide_ws_floats(B, V0 + E0) floats of scratch (shared with harp_mesh_chain_bwd_wide).  (V0 + E0) / 4 <= 1024 and (V0 + E0) * 12 B <= 64 KB. */
size_t harp_mesh_chain_wide_ws_floats(int B, int V);
int harp_mesh_chain_fwd_wide(const harp_mesh_chain* a, int clear_grads, float* part_ws, hipStream_t stream);
/* harp_mesh_chain_bwd on four workgroups per frame: projections backward + normal-length backward | normal gather + displacement |
 * normal gather | SubdivideMeshes backward — four launches, the same outputs (camera sums are sums of four partial sums).  Not for
 * a->light_only (HARP_ERR_ARG: that variant is one small pass of harp_mesh_chain_bwd).  (V0 + E0) * 24 B of LDS per workgroup. */
int harp_mesh_chain_bwd_wide(const harp_mesh_chain* a, float* part_ws, hipStream_t stream);

/* ---- losses, texture helpers, optimiser ---------------------------------------------------------------------------
 * Every loss call accumulates (+=) its value into `loss` and, if `w` (device pointer to the weight(s) = d total/d term)
 * and the gradient output are non-NULL, its weighted gradient (+= unless noted). */
/* torch.nn.L1Loss between pred*mask and target[fid]*mask[fid] (optimize_sequence.py:519, 543); g_pred is overwritten */
int harp_image_l1(const float* pred, const float* target, const float* mask, const int32_t* fid, int B, int n_per_frame, int C,
                  const float* w, float* loss, float* g_pred, hipStream_t stream);
/* kps_loss (loss/kps_loss.py:4-17); gt (T,21,3) mm indexed by fid, pred (B,n_joints_pred,3) m */
int harp_kps_loss(const float* gt, const int32_t* fid, const float* pred, int B, int n_joints_pred, const float* w, float* loss,
                  float* g_pred, hipStream_t stream);
/* loss[0..2], w[0..2] = mesh_laplacian_smoothing, mesh_normal_consistency (optimize_sequence.py:536-537),
 * arap_loss (loss/arap.py:4-57; ref_verts (V,3), NULL skips it); nbr_*: vertex->neighbour CSR (= the edge list, E edges),
 * nc_pairs (P,4) [v0,v1,a,b] per face pair sharing edge (v0,v1), vp_*: vertex -> (pair*4+role) CSR (harp_amd/topology.py) */
int harp_mesh_regularizers(const float* verts, const float* ref_verts, const int32_t* nbr_off, const int32_t* nbr_idx,
                           const int32_t* nc_pairs, const int32_t* vp_off, const int32_t* vp_idx, int B, int V, int P, int E,
                           const float* w, float* loss, float* g_verts, hipStream_t stream);
/* harp_mesh_regularizers + harp_kps_loss (kps_gt, fid, kps_pred, n_joints_pred, w_kps, loss_kps, g_kps_pred as there) in ONE launch:
 * both run between the hand layer and the backward pass of a fitting step and depend on nothing else (optimize_sequence.py:523-537) */
int harp_mesh_kps_terms(const float* verts, const float* ref_verts, const int32_t* nbr_off, const int32_t* nbr_idx,
                        const int32_t* nc_pairs, const int32_t* vp_off, const int32_t* vp_idx, int B, int V, int P, int E,
                        const float* w, float* loss, float* g_verts, const float* kps_gt, const int32_t* fid, const float* kps_pred,
                        int n_joints_pred, const float* w_kps, float* loss_kps, float* g_kps_pred, hipStream_t stream);
/* torch.sum(verts_disps ** 2) (optimize_sequence.py:533) */
int harp_sum_squares(const float* x, int n, const float* w, float* loss, float* g, hipStream_t stream);
/* torch.nn.MSELoss()(x, y) over n elements, ACCUMULATED into *loss (zero it first), and d/dx written to g_x (may be NULL): the
 * objective of the MANO-to-METRO vertex fit (metro_modifications/hand_utils.py:71, 80, 100) */
int harp_mse(const float* x, const float* y, int n, float* loss, float* g_x, hipStream_t stream);
/* albedo_reg / smooth_texture_reg (loss/texture_reg.py:5-30, 48-66) with the drawn integer offsets dist (H,W,2) */
int harp_texture_smooth_reg(const float* tex, const int32_t* dist, const float* mask, int H, int W, const float* w, float* loss,
                            float* g_tex, hipStream_t stream);
/* draws dist (H,W,2) = int(N(0,std)) like torch.normal(...).to(torch.int) at loss/texture_reg.py:15, 51, from a counter-based generator
 * (same values on every rank for the same seed) and, if dist2 != NULL, an independent second set with std2 in the same launch.
 * *counter_dev (one int, device memory) is advanced on the device after the draw, so graph replays draw fresh offsets. */
int harp_draw_texture_offsets(unsigned seed, int* counter_dev, int H, int W, float std, int32_t* dist, float std2, int32_t* dist2,
                              hipStream_t stream);
/* The parameter-only regularisers of a step in ONE launch (optimize_sequence.py:533, 549-551): harp_texture_smooth_reg(tex, dist_albedo)
 * -> loss_albedo / g_tex, harp_close_to_z_reg(nmap, z_scale) + harp_texture_smooth_reg(nmap, dist_normal) -> loss_normal / g_nmap (one
 * weight, as the reference adds the two into one term), and — when disp != NULL — harp_sum_squares(disp, n_disp) -> loss_disp / g_disp.
 * Same per-element arithmetic as the stand-alone calls; the gradient images are accumulated with atomics.  draw_counter_bump (optional):
 * *draw_counter_bump += 1 — the counter harp_step_prologue drew dist_albedo / dist_normal for, advanced by the launch that consumes them. */
int harp_texture_terms(const float* tex, const float* nmap, const float* mask, const int32_t* dist_albedo, const int32_t* dist_normal,
                       int H, int W, float z_scale, const float* w_albedo, float* loss_albedo, float* g_tex, const float* w_normal,
                       float* loss_normal, float* g_nmap, const float* disp, int n_disp, const float* w_disp, float* loss_disp,
                       float* g_disp, int* draw_counter_bump, hipStream_t stream);
/* scale * close_to_z_reg (loss/texture_reg.py:40-45) */
int harp_close_to_z_reg(const float* nm, int H, int W, float scale, const float* w, float* loss, float* g_nm, hipStream_t stream);
/* F.normalize(normal_map, dim=-1) (utils/visualize.py:99); n = number of texels */
int harp_normalize3_fwd(const float* x, int n, float* y, hipStream_t stream);
int harp_normalize3_bwd(const float* x, const float* gy, int n, float* gx, hipStream_t stream);
/* torch.optim.Adam step on a flat fp32 segment (optimize_sequence.py:265-310, 567-573); grad is read as g*grad_scale */
int harp_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                   float grad_scale, hipStream_t stream);

/* graph-replayable variant: hyper-parameters live in device memory (the host updates lr between epochs, e.g. for
 * ReduceLROnPlateau, optimize_sequence.py:309, 581-582); harp_adam_tick advances `step` and the bias corrections. */
typedef struct harp_adam_hyper {
  float lr, beta1, beta2, eps, grad_scale;
  int step;
  float step_size, inv_sqrt_bc2;   /* derived by harp_adam_tick */
} harp_adam_hyper;
int harp_adam_tick(harp_adam_hyper* h_dev, int count, hipStream_t stream);   /* `count` consecutive structs (one per param group) */
int harp_adam_apply(float* p, const float* g, float* m, float* v, size_t n, const harp_adam_hyper* h_dev, hipStream_t stream);
/* both optimisers of a stage (opt_coarse + opt_app, optimize_sequence.py:567-573) in one launch: elements [o0, o0+n0) of the four
 * arenas step with h2_dev[0], elements [o1, o1+n1) with h2_dev[1] */
int harp_adam_apply2(float* p, const float* g, float* m, float* v, size_t o0, size_t n0, size_t o1, size_t n1,
                     const harp_adam_hyper* h2_dev, hipStream_t stream);
/* The head of a fitting step's parameter-only work in ONE launch (any part optional): zero[0..n_zero) = 0 (the step's gradient slab),
 * harp_adam_tick(hyper, n_hyper), and the draw of harp_draw_texture_offsets for the CURRENT value of *draw_counter — which is NOT
 * advanced here (harp_step_frame.draw_counter: the epilogue of the same step advances it). */
int harp_step_prologue(float* zero, size_t n_zero, harp_adam_hyper* hyper_dev, int n_hyper, unsigned seed, const int* draw_counter,
                       int H, int W, float std, int32_t* dist, float std2, int32_t* dist2, hipStream_t stream);

/* ---- per-frame glue of the fitting loop ---------------------------------------------------------------------------
 * replaces the row gathers params[k][fid] of utils/visualize.py:26-27,38-39 / optimize_sequence.py:464, the camera
 * convention of utils/visualize.py:268-271, optimize_sequence.py:453-456 (shared light), :478-480 + renderer_helper.py:
 * 435-441 (ambient ratio -> light colours) and process_info_for_shadow (renderer_helper.py:454-468).
 * The tables are the reference's parameter dict (optimize_sequence.py:181-250) laid out in one flat fp32 arena. */
typedef struct harp_frame_tables {
  const float *pose, *rot, *trans, *cam;   /* (T,45) (T,3) (T,3) (T,3) */
  const float *shape;                      /* (10,) */
  const float *light_positions;            /* (T,3) */
  const float *amb_ratio;                  /* (1,) pre-sigmoid */
  float *g_pose, *g_rot, *g_trans, *g_cam, *g_shape, *g_light_positions, *g_amb_ratio;   /* gradient arena (+=), NULL = skip */
  int share_light;                         /* configs["share_light_position"] */
  /* SMPL-X arm path (use_arm): pose rows become [rot(3), wrist_pose(3), pose(45)] and betas are padded with zeros to n_betas_out */
  const float *wrist_pose;                 /* (T,3) or NULL (MANO path: rows are [rot(3), pose(45)]) */
  float *g_wrist_pose;
  int n_betas_out;                         /* 10 (MANO) or 20 (SMPL-X: 10 betas + 10 zero expression coefficients) */
} harp_frame_tables;
int harp_frame_setup_fwd(const harp_frame_tables* t, const int32_t* fid, int B, int S, float focal, int self_shadow, float* pose48,
                         float* betas, float* trans_b, float* cam_R, float* cam_T, float* light_pos, float* colors,
                         hipStream_t stream);
int harp_frame_setup_bwd(const harp_frame_tables* t, const int32_t* fid, int B, int S, float focal, int self_shadow,
                         const float* g_pose48, const float* g_betas, const float* g_trans_b, const float* g_cam_T,
                         const float* g_light_pos, const float* g_colors, hipStream_t stream);

/* ---- fused per-frame front of a fitting step, MANO path (csrc/hand_front.hip) ------------------------------------------
 * ONE launch = harp_frame_setup_fwd + harp_lbs_mano_fwd + harp_mesh_chain_fwd.  chain.verts_mm / chain.joints_mm are OUTPUTS
 * here; chain.cam_R / cam_T / light_pos must alias cam_R / cam_T / light_pos below (written by this call).  tables.wrist_pose
 * must be NULL (MANO rows), chain.V0 = 778, chain.NJ = 21. */
/* Optional book-keeping of a fitting step carried by the two per-frame launches instead of kernels of their own (all NULL / 0: none).
 * Front (harp_hand_front_fwd): with `schedule` set, workgroup b takes its frame from row (sched_row[0] mod n_rows) of the (n_rows,B)
 * device schedule and WRITES it to fid[b] (fid is then an output) and fid - target_offset to tfid_out[b] — the DataLoader's batch
 * (optimize_sequence.py:396-399, :446), what harp_schedule_next does as a launch; clear_mesh_grads: chain.g_vd / chain.g_joints_m of
 * the frame are zeroed (the two gradient segments the key-point / mesh terms accumulate into).
 * Back (harp_hand_back_bwd; every kernel that adds to the loss vector has finished by then): sched_row[0] = row + 1;
 * loss_out[0..n_loss) = loss[0..n_loss) and loss[..] = 0 (the terms of the NEXT step accumulate into a clean vector, no clear at the
 * head of a step), loss_total[0] += loss_w . loss when both are given; draw_counter[0] += 1 (harp_draw_texture_offsets_at of the next step draws fresh offsets). */
typedef struct harp_step_frame {
  const int32_t* schedule;
  const int32_t* tschedule;   /* optional (n_rows,B): rows of the resident targets (a dataset that holds a subset / another order of the frames);
                               * NULL: tfid = fid - target_offset */
  int32_t* sched_row;
  int n_rows, target_offset;
  int32_t* tfid_out;
  int clear_mesh_grads;
  float* loss;
  float* loss_out;
  int n_loss;
  int32_t* draw_counter;
  const float* loss_w;        /* optional (n_loss,): back: loss_total[0] += sum_k loss_w[k] * loss[k] — the step's sum_loss */
  float* loss_total;          /*   (optimize_sequence.py:553-559) accumulated over an epoch on the device, no per-step host arithmetic */
} harp_step_frame;

typedef struct harp_hand_front {
  harp_mesh_chain chain;
  harp_mano_model mano;
  harp_frame_tables tables;
  const int32_t* fid;        /* (B,) frame of each batch row */
  float *pose48, *betas, *trans_b, *cam_R, *cam_T, *light_pos, *colors;   /* as harp_frame_setup_fwd */
  float* lbs_ws;             /* harp_lbs_mano_ws_floats(B) floats, kept for harp_lbs_mano_bwd */
  int self_shadow;           /* colours from amb_ratio (shadow renderer) or the fixed Phong lights */
  harp_step_frame step;      /* optional prologue / epilogue of a fitting step (zero-initialised: none) */
} harp_hand_front;
int harp_hand_front_fwd(const harp_hand_front* h, hipStream_t stream);
/* harp_hand_front_fwd on four workgroups per frame: the hand layer with a quarter of the 778 vertices per workgroup (every workgroup
 * gathers the rows and runs the 16-joint chain; the frame's 1.35 MB of blend-shape rows go through four CUs' L1) + harp_mesh_chain_fwd_wide:
 * THREE launches of 4 B workgroups.  Same outputs, same `step` semantics.  part_ws as for harp_mesh_chain_fwd_wide. */
int harp_hand_front_wide_fwd(const harp_hand_front* h, float* part_ws, hipStream_t stream);
/* hybrid: the hand layer on four workgroups per frame, then harp_mesh_chain_fwd (one workgroup per frame): two launches, same outputs */
int harp_hand_front_hybrid_fwd(const harp_hand_front* h, hipStream_t stream);
/* The counterpart for the backward tail of a step (csrc/hand_back.hip): harp_mesh_chain_bwd + harp_lbs_mano_bwd + harp_frame_setup_bwd
 * (autograd of utils/visualize.py:16-88 / manopth/manolayer.py:108-296 down to the rows params[...][fid]) as THREE launches instead of
 * six: mesh chain + joint split + per-vertex skinning backward + trans / cam / light scatter per frame, the two vertex reductions, the
 * kinematic chain backward with the pose / rot / shape scatter.  `h` as passed to harp_hand_front_fwd of the same step, with the chain's
 * gradient inputs filled in (g_ndc_c, g_ndc_l, g_vd, g_n2, g_joints_m, g_light_R / g_light_T, has_normal_grad); gradients are ADDED to
 * the rows tables.g_* (pose, rot, trans, cam, shape and — when g_colors is given, i.e. an appearance stage ran — light_positions,
 * amb_ratio).  g_colors: 9 floats (dL/d colours) or NULL; g_betas_scratch: B*10 floats of scratch. */
int harp_hand_back_bwd(const harp_hand_front* h, const float* g_colors, float* g_betas_scratch, hipStream_t stream);
/* harp_hand_back_bwd with the mesh-chain backward (harp_mesh_chain_bwd_wide's first three launches) and the per-vertex hand-layer backward on
 * four workgroups per frame: SIX launches of mostly 4 B workgroups instead of three of B.  Same results to float32 summation order;
 * chain.light_only is forwarded to harp_hand_back_bwd.  part_ws as for harp_mesh_chain_fwd_wide. */
int harp_hand_back_wide_bwd(const harp_hand_front* h, const float* g_colors, float* g_betas_scratch, float* part_ws, hipStream_t stream);

/* ---- fused per-frame front / back of a fitting step, SMPL-X arm path (csrc/arm_front.hip) ---------------------------------
 * replaces, for configs["use_arm"], what harp_hand_front_fwd / harp_hand_back_bwd replace for the MANO hand: the row gathers
 * params[...][fid] (utils/visualize.py:26-27, 37-40), SMPLXARM.forward(..., return_type='mano_w_arm') (hand_models_harp/
 * body_models.py:2163-2390) and prepare_mesh (utils/visualize.py:45-64) with both projections and the light camera
 * (renderer_helper.py:344, 353, 454-468) — and their autograd down to the gradient rows of the parameter tables.
 * Front = THREE launches: harp_frame_setup_fwd + joint chain per frame | the blend-shape contraction of all frames on the matrix
 * cores | skinning + output joints + harp_mesh_chain_fwd per frame.  Back = FOUR: harp_mesh_chain_bwd + joint split + recentring
 * + skinning backward + trans / cam / light / ambient scatter per frame | the two vertex reductions (MFMA) | the kinematic chain
 * backward with the rot / wrist_pose / pose / shape scatter.  Same arithmetic as harp_frame_setup_* + harp_lbs_tree_* +
 * harp_mesh_chain_*; `step` as in harp_hand_front.  tables.wrist_pose must be set (rows [rot, wrist_pose, pose] = 51 floats),
 * tables.n_betas_out = tree.NB, chain.V0 = tree.NV, chain.NJ = tree.n_joints_out; chain.verts_mm / joints_mm are OUTPUTS;
 * chain.cam_R / cam_T / light_pos must alias cam_R / cam_T / light_pos below. */
typedef struct harp_arm_front {
  harp_mesh_chain chain;
  harp_tree_model tree;
  harp_frame_tables tables;
  const int32_t* fid;        /* (B,) frame of each batch row */
  float *pose_in, *betas, *trans_b, *cam_R, *cam_T, *light_pos, *colors;   /* as harp_frame_setup_fwd: (B,51) (B,NB) (B,3) (B,9) (B,3) (B,3) (9) */
  float* lbs_ws;             /* harp_lbs_tree_ws_floats(&tree, B) floats, shared by the front and the back of one step */
  const float* weights_T;    /* (NJ, NV): tree.weights transposed (the per-frame kernels read one coalesced row per joint) */
  int self_shadow;           /* colours from amb_ratio (shadow renderer) or the fixed Phong lights */
  harp_step_frame step;      /* optional prologue / epilogue of a fitting step (zero-initialised: none) */
} harp_arm_front;
int harp_arm_front_fwd(const harp_arm_front* h, hipStream_t stream);
/* `h` as passed to harp_arm_front_fwd of the same step, with the chain's gradient inputs filled in (see harp_hand_back_bwd); gradients
 * are ADDED to the rows tables.g_* (pose, rot, wrist_pose, trans, cam, shape and — when g_colors is given — light_positions, amb_ratio).
 * g_pose_scratch: B*51 floats (holds dL/d pose rows afterwards), g_betas_scratch: B*NB floats. */
int harp_arm_back_bwd(const harp_arm_front* h, const float* g_colors, float* g_pose_scratch, float* g_betas_scratch, hipStream_t stream);
/* The same front / back with the per-frame work on FOUR workgroups per frame around harp_mesh_chain_fwd_wide / _bwd_wide
 * (a quarter of the vertices per workgroup; 5 + 7 launches).  part_ws: harp_mesh_chain_wide_ws_floats(B, V0 + E0) floats.
 * chain.light_only is forwarded to harp_arm_back_bwd. */
int harp_arm_front_wide_fwd(const harp_arm_front* h, float* part_ws, hipStream_t stream);
int harp_arm_back_wide_bwd(const harp_arm_front* h, const float* g_colors, float* g_pose_scratch, float* g_betas_scratch, float* part_ws,
                           hipStream_t stream);

int harp_light_setup_fwd(const float* centroid, const float* light_pos, int B, float* light_R, float* light_T, hipStream_t stream);
int harp_light_setup_bwd(const float* centroid, const float* light_pos, const float* g_light_R, const float* g_light_T, int B, int V,
                         float* g_light_pos, float* g_centroid, float* g_verts, hipStream_t stream);
int harp_scale(const float* x, float s, int n, float* y, hipStream_t stream);
/* replaces the DataLoader's batch of frame ids (optimize_sequence.py:396-399, :446): row (counter[0] mod n_rows) of a device-resident
 * (n_rows,B) int32 schedule -> fid (B,), tfid = fid - target_offset; then counter[0] = row + 1.  Graph-replayable. */
int harp_schedule_next(const int32_t* schedule, int n_rows, int B, int target_offset, int32_t* counter, int32_t* fid, int32_t* tfid,
                       float* zero, int n_zero, hipStream_t stream);   /* zero (optional): n_zero floats cleared in the same launch (loss vector) */
/* the same with the target rows given by a second (n_rows,B) table instead of fid - target_offset (tschedule == NULL: as above) */
int harp_schedule_next_rows(const int32_t* schedule, const int32_t* tschedule, int n_rows, int B, int target_offset, int32_t* counter,
                            int32_t* fid, int32_t* tfid, float* zero, int n_zero, hipStream_t stream);

/* ---- perceptual term: 3x3 convolutions on the matrix cores ----------------------------------------------------------------------------
 * replaces the torch.nn.Conv2d / ReLU / MaxPool2d stack of model/vgg.py:10-56 (torchvision vgg16.features[0:23]; built at
 * optimize_sequence.py:405, evaluated on y_pred * mask and y_true * mask at :546-547, weight 1.0 at :419) and its autograd: F.conv2d
 * (cuDNN in the reference), threshold_backward, max_pool2d_with_indices_backward, L1Loss.  The filters are frozen (requires_grad=False,
 * model/vgg.py:34-36): only data gradients exist.
 * Activations are NHWC float32.  harp_conv3x3 is one 3x3 / pad 1 / stride 1 convolution (Cin % 16 == 0, Cout % 64 == 0; pad the
 * channels with zeros otherwise) with one of four fused epilogues:
 *   HARP_CONV_RELU      out = relu(conv + bias); pooled (optional, H and W even) = max_pool2d(out, 2, 2); out may be NULL when pooled is not
 *   HARP_CONV_RELU_TAP  the same, plus the tap's share of L1Loss(features(pred), features(target)): *loss (+=, double) tap_scale * sum|out - target|,
 *                       g_tap = tap_scale * sign(out - target) * [out > 0]  (d loss / d conv, ReLU backward applied);
 *                       target (T,H,W,Cout) row target_row[n] (NULL: row n)
 *   HARP_CONV_GATE      data gradient through the ReLU in front of this convolution's input: out = conv * [gate > 0], gate (N,H,W,Cout)
 *   HARP_CONV_UNPOOL    data gradient through max pool + ReLU: the convolution runs at the pooled size (H,W); out and gate are (N,2H,2W,Cout);
 *                       out (+=) the result routed to the first maximum of each 2x2 window of gate where that maximum is positive
 * precision 0: v_mfma_f32_32x32x2_f32 (float32 fma chain); 1: three-term bf16 split, float32 accumulate (~16 mantissa bits per product).
 * filters: harp_conv3x3_pack_filters output for the same precision (transpose = 1 packs the data-gradient filters of a forward (Cout,Cin,3,3)
 * weight: channels swap roles, taps are mirrored). */
#define HARP_CONV_RELU 0
#define HARP_CONV_RELU_TAP 1
#define HARP_CONV_GATE 2
#define HARP_CONV_UNPOOL 3
typedef struct harp_conv3x3_args {
  const float* in;            /* (N,H,W,Cin) */
  const void* filters;        /* harp_conv3x3_filter_bytes(Cout,Cin) bytes */
  const float* bias;          /* (Cout) or NULL */
  float* out;
  float* pooled;              /* (N,H/2,W/2,Cout) or NULL */
  const float* target;
  const int32_t* target_row;
  float* g_tap;
  double* loss;
  const float* gate;
  int N, H, W, Cin, Cout;
  int precision, epilogue;
  int in_channels;            /* channels per pixel of `in` in memory (multiple of 4, <= Cin; the rest reads as zero); 0 = Cin */
  float tap_scale;
  /* bounded mode (all NULL / 0 = every tile): image n belongs to frame row r = target_row[n]; only the 16x16 output tiles of that frame's
   * list are computed, and input pixels in cells this pass did not write are taken from in_alt (the same activation of another pass,
   * e.g. the target frame's, rows through target_row) or read as zero */
  const int32_t* tile_list;   /* (T, max_tiles): ty * tile_pitch + tx in the frame's tile grid */
  const int32_t* tile_count;  /* (T) */
  int max_tiles;
  int in_valid_shift;         /* cells of in_valid are (8 << shift) input pixels square: 1 = the producer's 16x16 tiles, 0 = tiles of a 2x finer producer behind a pool */
  const int32_t* in_valid;    /* (T, in_valid_pitch^2) 0/1 over the PRODUCER's tile grid */
  const float* in_alt;        /* (T,H,W,in_channels) or NULL: zero */
  const int32_t* out_valid;   /* HARP_CONV_UNPOOL: (T, out_valid_pitch^2) 0/1 over the tile grid of `out` (twice this convolution's resolution); windows elsewhere are skipped */
  /* a frame's tile grid may be shifted so that its tiles hug the active region: tile (ty, tx) covers pixels [16 ty - oy, 16 ty - oy + 16) x
   * [16 tx - ox, ...), origin (oy, ox) EVEN (2x2 pool windows stay inside a tile), per frame, in the pixels of the grid's own resolution;
   * NULL = (0, 0).  pitch = tiles per row (and rows) of the grid's bitmap */
  const int32_t* tile_origin;       /* (T,2) of the grid tile_list indexes */
  const int32_t* in_valid_origin;   /* (T,2) of the producer's grid */
  const int32_t* out_valid_origin;  /* (T,2) of `out`'s grid */
  int tile_pitch, in_valid_pitch, out_valid_pitch;
  /* tiles of 8x8 pixels (the coarse levels of the perceptual term, where a hand is a few tiles across): tile_side = 8 — tile_list / tile_origin /
   * tile_pitch then describe a grid of 8-pixel tiles and a workgroup computes four of them (one per wave); 0 or 16 = the 16x16 form.
   * in_valid_cell: side of in_valid's cells in INPUT pixels when the producer's tiles are not 16 pixels (4 — only with tile_side 8 —, 8, 16;
   * 0 = 8 << in_valid_shift; in_valid_shift still says whether the producer sits behind a pool).  out_valid_cell: side of `out`'s tiles (8 or 16; 0 = 16). */
  int tile_side, in_valid_cell, out_valid_cell;
} harp_conv3x3_args;
size_t harp_conv3x3_filter_bytes(int Cout, int Cin);
int harp_conv3x3_pack_filters(const float* w, int Cout, int Cin, int transpose, int precision, void* packed, hipStream_t stream);
int harp_conv3x3(const harp_conv3x3_args* a, hipStream_t stream);

/* The whole term (optimize_sequence.py:546-547): loss = L1Loss(vgg(y_pred * mask), vgg(y_true * mask)) with vgg = model/vgg.py's
 * Vgg16Features (rows: the image itself, relu1_2, relu2_2, relu3_3, relu4_3, scaled by layers_weights, :51-55), and d loss / d y_pred.
 * harp_vgg16: the ten convolutions in the order of torchvision vgg16.features[0:23] (0, 2, 5, 7, 10, 12, 14, 17, 19, 21): filters[k] /
 * filters_t[k] = harp_conv3x3_pack_filters(w_k, transpose = 0 / 1) for `precision` (the first layer's 3 input channels padded to 16;
 * filters_t[0] unused), bias[k], w0t (9,3,64) = the first layer's data-gradient filters for the vector-ALU kernel that ends the
 * backward pass: w0t[t][c][co] = w0[co][c][8 - t] with w0 the (64,3,3,3) weight, taps flattened.
 * ws: harp_vgg16_ws_bytes(N,S,with_gradient) bytes (S % 8 == 0), zero-filled once by the caller before the first use.
 * harp_vgg16_features: activations of image[rows[n]] * mask[rows[n]] (rows NULL: n), NHWC, into the caller's arrays — its cache of the
 *   target frames' features (they do not change during a fit): out[k], k = 0..9 = relu(convolution k) (N, S/d_k, S/d_k, Cout_k) with
 *   d = 1,1,2,2,4,4,4,8,8,8; out[10..12] = the three pooled maps (N,S/2,S/2,64), (N,S/4,S/4,128), (N,S/8,S/8,256).  The four taps
 *   out[1], out[3], out[6], out[9] (relu1_2 ... relu4_3) are required, the others may be NULL (kept in ws only).
 * Bounded mode of harp_vgg16_term (tiles[0] != NULL; needs target_by_row and the cache of ALL activations): the stack runs only in the
 *   tiles (tile_side[L]: 16 pixels a side, or 8) of each resolution level (L = 0..3, side S >> L) where the rendered image's activations can differ from the target
 *   frame's — tiles[L] (T, tile_pitch[L]^2) 0/1 per frame over a tile grid shifted by the frame's tile_origin[L] (so that the tiles hug the
 *   active region), tile_list[L] (T, max_tiles[L]) their indices, tile_count[L] (T): the support
 *   of mask grown by the receptive field (the caller's set-up, harp_amd/model/vgg_hip.py).  Elsewhere pred == target exactly: the L1
 *   and its gradient vanish, and inputs needed from there are read from target_in[k] = the target frame's input activation of
 *   convolution k (k = 1..9: out[0], out[10], out[2], out[11], out[4], out[5], out[12], out[7], out[8] of harp_vgg16_features).  Same
 *   loss and gradient as the full pass, bit for bit in the tiles it computes.
 * harp_vgg16_term: forward over rgb * mask[rows[n]] with the L1 against target[k] fused into the tap layers, backward to the image:
 *   *loss = the term (unweighted);  g_rgb (N,S,S,3) = covered < 0 ? 0 : g_rgb + weight * d loss / d rgb   (covered NULL: everywhere). */
typedef struct harp_vgg16 {
  const void* filters[10];
  const void* filters_t[10];
  const float* bias[10];
  const float* w0t;
  float layer_w[5];
  int precision;
} harp_vgg16;
typedef struct harp_vgg16_term_args {
  const float* rgb;            /* (N,S,S,3) rendered images */
  const float* y_true;         /* (T,S,S,3) */
  const float* mask;           /* (T,S,S) */
  const int32_t* rows;         /* (N,) target row of each image; NULL: n */
  const float* target[4];      /* tap features of y_true * mask, layouts of harp_vgg16_features */
  int target_by_row;           /* 1: target[k] holds T rows indexed through `rows`; 0: N rows in batch order */
  const int32_t* covered;      /* (N,S,S) nearest-face ids (< 0: no face) or NULL */
  float* g_rgb;
  float weight;
  float* loss;
  int N, S;
  void* ws;
  const float* target_in[10];
  const int32_t* tiles[4];
  const int32_t* tile_list[4];
  const int32_t* tile_count[4];
  int max_tiles[4];
  const int32_t* tile_origin[4];   /* (T,2) even (oy, ox) per frame and level: tile (ty, tx) covers pixels [16 ty - oy, +16) x [16 tx - ox, +16) */
  int tile_pitch[4];               /* tiles per row (= rows) of tiles[L]; tile_list[L] entries are ty * pitch + tx */
  int tile_side[4];                /* side of level L's tiles in pixels: 16 (0 = 16) or 8; 16 at level 0, and never 16 behind (coarser than) an 8 */
  void* side_streams[3];           /* up to three more hipStream_t (NULL-terminated).  With k of them (and rows, target_by_row) the term runs as k + 1
                                    * parts of the batch, part i > 0 on side_streams[i - 1], forked from and joined back into `stream` by events
                                    * (capturable): the chains of 21 dependent launches fill each other's partly filled last rounds of workgroups.
                                    * Same results. */
} harp_vgg16_term_args;
size_t harp_vgg16_ws_bytes(int N, int S, int with_gradient);
int harp_vgg16_features(const harp_vgg16* net, const float* image, const float* mask, const int32_t* rows, int N, int S, void* ws,
                        float* const* out, hipStream_t stream);
int harp_vgg16_term(const harp_vgg16* net, const harp_vgg16_term_args* t, hipStream_t stream);

/* ---- data-parallel exchange (RCCL over xGMI) -------------------------------------------------------------------------
 * New capability: the reference is single-device.  Frames of a sequence are sharded over the GPUs of a node; between
 * `sum_loss.backward()` and `opt_coarse.step() / opt_app.step()` (optimize_sequence.py:567-573) every rank sums the flat fp32
 * gradient bucket [pose .. normal_map] of the parameter arena in place (1/world is applied by the Adam kernel's grad_scale).
 * RCCL is bound with dlopen at the first harp_comm_* call (the copy an embedding PyTorch process has mapped is reused).
 * harp_comm_unique_id: rank 0 obtains HARP_COMM_ID_BYTES opaque bytes and hands them to every rank over any side channel;
 * harp_comm_create: collective over all ranks, binds the calling thread's current HIP device; the handle is opaque.
 * harp_allreduce_flat only enqueues on `stream` (no host synchronisation), so it can be captured into a hipGraph. */
#define HARP_COMM_ID_BYTES 128
#define HARP_ERR_NO_RCCL 1000
#define HARP_ERR_COMM 1001
int harp_comm_unique_id(void* id_out);
int harp_comm_create(const void* id, int rank, int world, void** comm_out);
int harp_comm_destroy(void* comm);
int harp_allreduce_flat(void* comm, float* buf, size_t n, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HARP_HIP_H */
