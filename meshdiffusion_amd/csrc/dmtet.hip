// Marching tetrahedra on a static tet grid, one workgroup (16 wavefronts) per mesh.
//
// Reference: nvdiffrec/lib/geometry/dmtet.py:105-163 (DMTet.__call__), LUTs :34-54.
// The reference sorts+uniques the edges of the valid tets at run time (torch.unique(dim=0));
// because the tet grid is static we precompute ONCE the lexicographically sorted unique edge
// list of all tets and the tet->edge-id table.  A crossing edge (exactly one endpoint with
// sdf > 0) always belongs to a valid tet, so numbering the crossing edges by an exclusive
// prefix sum over that static sorted order reproduces the reference's vertex ids exactly;
// faces are emitted 1-triangle tets first (tet order), then 2-triangle tets, like the
// reference's torch.cat.  Prefix sums are wave-level ballots + popcounts (wave64).
#include "md_common.h"

#pragma clang fp contract(off)

static constexpr int MT_THREADS = 1024;
static constexpr int MT_WAVES = MT_THREADS / 64;

__device__ __constant__ int8_t c_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
__device__ __constant__ int8_t c_num_tri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};

// exclusive prefix of a 0/1 flag over the 1024-thread block; returns block total via `total`.
__device__ __forceinline__ int block_excl_scan_flag(bool flag, int* lds_wave, int& total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  const int wprefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) lds_wave[wid] = __popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MT_WAVES; ++w) {
    const int c = lds_wave[w];
    if (w < wid) base += c;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return base + wprefix;
}

__global__ __launch_bounds__(MT_THREADS) void md_marching_tets_kernel(
    const float* __restrict__ pos, const float* __restrict__ sdf, const int32_t* __restrict__ tets,
    const int32_t* __restrict__ edges, const int32_t* __restrict__ tet_edges, int n_verts, int n_edges,
    int n_tets, float* __restrict__ verts, int64_t* __restrict__ faces, int64_t* __restrict__ face_tet,
    int32_t* __restrict__ counts, int32_t* __restrict__ workspace) {
  __shared__ int lds_wave[MT_WAVES];
  __shared__ int lds_red[2][MT_WAVES];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float* mpos = pos + (int64_t)m * n_verts * 3;
  const float* msdf = sdf + (int64_t)m * n_verts;
  float* mverts = verts + (int64_t)m * n_edges * 3;
  int64_t* mfaces = faces + (int64_t)m * n_tets * 2 * 3;
  int64_t* mftet = face_tet ? face_tet + (int64_t)m * n_tets * 2 : nullptr;
  int32_t* vid = workspace + (int64_t)m * n_edges;  // edge -> vertex id or -1

  // ---- phase 1: crossing edges -> vertex ids + interpolated positions ----
  int vbase = 0;
  for (int e0 = 0; e0 < n_edges; e0 += MT_THREADS) {
    const int e = e0 + tid;
    bool cross = false;
    int a = 0, b = 0;
    float sa = 0.f, sb = 0.f;
    if (e < n_edges) {
      a = edges[2 * e]; b = edges[2 * e + 1];
      sa = msdf[a]; sb = msdf[b];
      cross = (sa > 0.f) != (sb > 0.f);
    }
    int tot;
    const int pre = block_excl_scan_flag(cross, lds_wave, tot);
    if (e < n_edges) {
      if (cross) {
        const int v = vbase + pre;
        vid[e] = v;
        // reference: sdf pair (s0, -s1); w = flip(pair)/sum(pair); vert = p0*w0 + p1*w1
        const float nsb = -sb;
        const float den = sa + nsb;
        const float w0 = nsb / den, w1 = sa / den;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float t0 = mpos[a * 3 + k] * w0;
          const float t1 = mpos[b * 3 + k] * w1;
          mverts[(int64_t)v * 3 + k] = t0 + t1;
        }
      } else {
        vid[e] = -1;
      }
    }
    vbase += tot;
  }
  __threadfence_block();
  __syncthreads();

  // ---- phase 2a: count 1- and 2-triangle tets ----
  int c1 = 0, c2 = 0;
  for (int t = tid; t < n_tets; t += MT_THREADS) {
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) idx |= (msdf[tets[4 * t + k]] > 0.f) ? (1 << k) : 0;
    const int nt = c_num_tri[idx];
    c1 += (nt == 1);
    c2 += (nt == 2);
  }
  {
    int s1 = c1, s2 = c2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((tid & 63) == 0) { lds_red[0][tid >> 6] = s1; lds_red[1][tid >> 6] = s2; }
  }
  __syncthreads();
  int N1 = 0, N2 = 0;
#pragma unroll
  for (int w = 0; w < MT_WAVES; ++w) { N1 += lds_red[0][w]; N2 += lds_red[1][w]; }
  __syncthreads();

  // ---- phase 2b: emit faces (1-tri tets first, then 2-tri tets, both in tet order) ----
  int b1 = 0, b2 = 0;
  for (int t0 = 0; t0 < n_tets; t0 += MT_THREADS) {
    const int t = t0 + tid;
    int idx = 0, nt = 0;
    if (t < n_tets) {
#pragma unroll
      for (int k = 0; k < 4; ++k) idx |= (msdf[tets[4 * t + k]] > 0.f) ? (1 << k) : 0;
      nt = c_num_tri[idx];
    }
    int tot1, tot2;
    const int p1 = block_excl_scan_flag(nt == 1, lds_wave, tot1);
    const int p2 = block_excl_scan_flag(nt == 2, lds_wave, tot2);
    if (nt > 0) {
      int ev[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ev[k] = vid[tet_edges[6 * t + k]];
      const int64_t f0 = (nt == 1) ? (int64_t)(b1 + p1) : (int64_t)N1 + 2 * (int64_t)(b2 + p2);
      for (int f = 0; f < nt; ++f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int ei = c_tri_table[idx][f * 3 + k];
          int v = 0;
#pragma unroll
          for (int q = 0; q < 6; ++q) v = (ei == q) ? ev[q] : v;
          mfaces[(f0 + f) * 3 + k] = (int64_t)v;
        }
        if (mftet) mftet[f0 + f] = (int64_t)t;
      }
    }
    b1 += tot1;
    b2 += tot2;
  }
  if (tid == 0) {
    counts[m * 4 + 0] = vbase;
    counts[m * 4 + 1] = N1 + 2 * N2;
    counts[m * 4 + 2] = N1;
    counts[m * 4 + 3] = N2;
  }
}

extern "C" int64_t md_marching_tets_workspace_bytes(int32_t n_meshes, int32_t n_edges) {
  if (n_meshes <= 0 || n_edges <= 0) return MD_ERR_BAD_ARG;
  return (int64_t)n_meshes * n_edges * 4;
}

extern "C" int md_marching_tets(const float* pos, const float* sdf, const int32_t* tets,
                                const int32_t* edges, const int32_t* tet_edges, int32_t n_meshes,
                                int32_t n_verts, int32_t n_edges, int32_t n_tets, float* verts,
                                int64_t* faces, int64_t* face_tet, int32_t* counts, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  if (!pos || !sdf || !tets || !edges || !tet_edges || !verts || !faces || !counts || !workspace ||
      n_meshes <= 0 || n_verts <= 0 || n_edges <= 0 || n_tets <= 0)
    return MD_ERR_BAD_ARG;
  if (workspace_bytes < md_marching_tets_workspace_bytes(n_meshes, n_edges)) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_marching_tets_kernel, dim3((unsigned)n_meshes), dim3(MT_THREADS), 0,
                     (hipStream_t)stream, pos, sdf, tets, edges, tet_edges, n_verts, n_edges, n_tets,
                     verts, faces, face_tet, counts, (int32_t*)workspace);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- smooth vertex normals of an extracted mesh (nvdiffrec/lib/render/mesh.py:200-229 `auto_normals`) ----------------
// f_nrm = cross(v1 - v0, v2 - v0) per face, splatted onto its three vertices (scatter_add in the reference: float
// atomics here, so the summation order -- not the set of terms -- differs), then v / sqrt(max(v.v, 1e-20)) with
// degenerate sums (v.v <= 1e-20) replaced by (0, 0, 1) first (util.safe_normalize, util.py:33-35).
__global__ void md_face_normals_kernel(const float* __restrict__ verts, const int64_t* __restrict__ faces, int64_t F,
                                       float* __restrict__ v_nrm, float* __restrict__ f_nrm) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int64_t i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
  const float ax = verts[i1 * 3] - verts[i0 * 3], ay = verts[i1 * 3 + 1] - verts[i0 * 3 + 1], az = verts[i1 * 3 + 2] - verts[i0 * 3 + 2];
  const float bx = verts[i2 * 3] - verts[i0 * 3], by = verts[i2 * 3 + 1] - verts[i0 * 3 + 1], bz = verts[i2 * 3 + 2] - verts[i0 * 3 + 2];
  const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
  if (f_nrm) { f_nrm[f * 3] = nx; f_nrm[f * 3 + 1] = ny; f_nrm[f * 3 + 2] = nz; }
  const int64_t idx[3] = {i0, i1, i2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    atomicAdd(v_nrm + idx[k] * 3, nx); atomicAdd(v_nrm + idx[k] * 3 + 1, ny); atomicAdd(v_nrm + idx[k] * 3 + 2, nz);
  }
}

__global__ void md_normalize_normals_kernel(float* __restrict__ v_nrm, int64_t V) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float x = v_nrm[v * 3], y = v_nrm[v * 3 + 1], z = v_nrm[v * 3 + 2];
  float d = x * x + y * y + z * z;
  if (!(d > 1e-20f)) { x = 0.f; y = 0.f; z = 1.f; d = 1.f; }
  const float len = sqrtf(fmaxf(d, 1e-20f));
  v_nrm[v * 3] = x / len; v_nrm[v * 3 + 1] = y / len; v_nrm[v * 3 + 2] = z / len;
}

extern "C" int md_vertex_normals(const float* verts, const int64_t* faces, int64_t n_verts, int64_t n_faces, float* v_nrm,
                                 float* f_nrm, void* stream) {
  if (!verts || !faces || !v_nrm || n_verts <= 0 || n_faces < 0) return MD_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(v_nrm, 0, (size_t)n_verts * 3 * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MD_HIP_CLEAR_ERROR();
  if (n_faces > 0)
    hipLaunchKernelGGL(md_face_normals_kernel, dim3((unsigned)((n_faces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts,
                       faces, n_faces, v_nrm, f_nrm);
  hipLaunchKernelGGL(md_normalize_normals_kernel, dim3((unsigned)((n_verts + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     v_nrm, n_verts);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
