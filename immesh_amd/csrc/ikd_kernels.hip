// Kernels of the legacy registration path: device point map + "Old map ICP" matcher (src/voxel_mapping.cpp:1400-1480),
// esti_plane (include/common_lib.h:356-402) and the H rows of :1487-1575 with R_inv = 1 / LASER_POINT_COV.  See ikd_map.hpp.
#include "ikd_map.hpp"
#include "dev_math.hpp"
#include "prof.hpp"
using namespace imd;
IMD int ik_sym21(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }   // r <= c, upper triangle of a 6 x 6, row-major

#define IK_BIAS (1 << 20)
#define IK_MASK ((1ull << 21) - 1)
#define IK_EMPTY 0xFFFFFFFFFFFFFFFFull

IMD unsigned long long ik_key(long x, long y, long z) {
    return (((unsigned long long)(x + IK_BIAS) & IK_MASK) << 42) | (((unsigned long long)(y + IK_BIAS) & IK_MASK) << 21) | ((unsigned long long)(z + IK_BIAS) & IK_MASK);
}
IMD long ik_cell(float v, float ds) { return (long)floorf(v / ds); }   // floor(PointToAdd.x / downsample_size), ikd_Tree.cpp:514
IMD float ik_dist(float ax, float ay, float az, float bx, float by, float bz) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz); }   // calc_dist :1722
IMD long long ik_find(const IkdMapDev& m, unsigned long long key) {
    unsigned long long h = hash64(key) & m.mask;
    for (int probe = 0; probe < 4096; probe++) {
        const unsigned long long k = m.keys[h];
        if (k == key) return (long long)h;
        if (k == IK_EMPTY) return -1;
        h = (h + 1) & m.mask;
    }
    return -1;
}
IMD long long ik_find_or_insert(const IkdMapDev& m, unsigned long long key) {
    unsigned long long h = hash64(key) & m.mask;
    for (int probe = 0; probe < 4096; probe++) {
        const unsigned long long k = atomicCAS(&m.keys[h], IK_EMPTY, key);
        if (k == IK_EMPTY || k == key) return (long long)h;
        h = (h + 1) & m.mask;
    }
    m.counters[2] = 1;
    return -1;
}

// KD_TREE::Build: every point is kept
__global__ void ikd_build_kernel(IkdMapDev m, const float* __restrict__ xyz, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
    const long long s = ik_find_or_insert(m, ik_key(ik_cell(x, m.ds), ik_cell(y, m.ds), ik_cell(z, m.ds)));
    if (s < 0) return;
    const int pos = atomicAdd(&m.count[s], 1);
    if (pos >= IKD_CELL_PTS) { atomicSub(&m.count[s], 1); m.counters[2] = 2; return; }
    m.pts[(size_t)s * IKD_CELL_PTS + pos] = make_float4(x, y, z, __int_as_float(i));
    atomicAdd(&m.counters[0], 1);
}

// Add_Points(.., downsample_on = true), ikd_Tree.cpp:493-545, one batch.  The sequential loop keeps, per downsample box, the point nearest
// to the box centre; a new point replaces the survivor when it is at least as near (stored points win only if strictly nearer, :527-533).
// Phase 0 marks the batch's cells, phase 1 takes the minimum over the batch's points (later index wins ties), phase 2 settles each cell.
__global__ void ikd_add_mark_kernel(IkdMapDev m, const float* __restrict__ xyz, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
    const long long s = ik_find_or_insert(m, ik_key(ik_cell(x, m.ds), ik_cell(y, m.ds), ik_cell(z, m.ds)));
    if (s < 0) return;
    if (atomicExch(&m.stamp[s], m.seq) != m.seq) { m.best[s] = ~0ull; m.touched[atomicAdd(&m.counters[1], 1)] = (int)s; }
}
IMD void ik_mid(float p, float ds, float* mid) {
    const float vmin = (float)(floor((double)(p / ds)) * (double)ds);
    const float vmax = vmin + ds;
    *mid = (float)((double)vmin + (double)(vmax - vmin) / 2.0);
}
__global__ void ikd_add_min_kernel(IkdMapDev m, const float* __restrict__ xyz, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
    const long long s = ik_find(m, ik_key(ik_cell(x, m.ds), ik_cell(y, m.ds), ik_cell(z, m.ds)));
    if (s < 0) return;
    float mx, my, mz;
    ik_mid(x, m.ds, &mx); ik_mid(y, m.ds, &my); ik_mid(z, m.ds, &mz);
    const float d = ik_dist(x, y, z, mx, my, mz);
    atomicMin(&m.best[s], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i));
}
__global__ void ikd_add_settle_kernel(IkdMapDev m, const float* __restrict__ xyz) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m.counters[1]) return;
    const int s = m.touched[k];
    const unsigned long long b = m.best[s];
    const int i = (int)(0xFFFFFFFFu - (unsigned int)(b & 0xFFFFFFFFull));
    const float dn = __uint_as_float((unsigned int)(b >> 32));
    const float x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
    float mx, my, mz;
    ik_mid(x, m.ds, &mx); ik_mid(y, m.ds, &my); ik_mid(z, m.ds, &mz);
    const int cnt = m.count[s];
    float4* cell = m.pts + (size_t)s * IKD_CELL_PTS;
    int bo = -1;
    float dbest = 3.4e38f;
    for (int j = 0; j < cnt; j++) {   // the stored point that the batch would have to beat: the first strictly nearest one
        const float4 q = cell[j];
        const float d = ik_dist(q.x, q.y, q.z, mx, my, mz);
        if (d < dbest) { dbest = d; bo = j; }
    }
    const bool new_wins = bo < 0 || dn <= dbest;
    if (cnt > 1 || new_wins) {   // Downsample_Storage.size() > 1 || same_point(new, result): delete the box, add the survivor
        const float4 keep = new_wins ? make_float4(x, y, z, __int_as_float(atomicAdd(&m.counters[3], 1))) : make_float4(cell[bo].x, cell[bo].y, cell[bo].z, __int_as_float(atomicAdd(&m.counters[3], 1)));
        cell[0] = keep;
        m.count[s] = 1;
        atomicAdd(&m.counters[0], 1 - cnt);
    }
}
// KD_TREE::Delete_Point_Boxes (ikd_Tree.cpp:655-690): one thread per hash slot compacts its cell
__global__ void ikd_delete_boxes_kernel(IkdMapDev m, IkdBoxes bx, int32_t* __restrict__ n_deleted) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > (long long)m.mask || m.keys[s] == IK_EMPTY) return;
    const int cnt = m.count[s];
    float4* cell = m.pts + (size_t)s * IKD_CELL_PTS;
    int w = 0;
    for (int j = 0; j < cnt; j++) {
        const float4 q = cell[j];
        bool in = false;
        for (int b = 0; b < bx.n && !in; b++)
            in = bx.b[b][0] <= q.x && bx.b[b][3] > q.x && bx.b[b][1] <= q.y && bx.b[b][4] > q.y && bx.b[b][2] <= q.z && bx.b[b][5] > q.z;
        if (!in) cell[w++] = q;
    }
    if (w != cnt) { m.count[s] = w; atomicAdd(&m.counters[0], w - cnt); atomicAdd(n_deleted, cnt - w); }
}
__global__ void ikd_dump_kernel(IkdMapDev m, float* __restrict__ xyz, long long cap, unsigned long long* __restrict__ count) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > (long long)m.mask || m.keys[s] == IK_EMPTY) return;
    const int cnt = m.count[s];
    for (int j = 0; j < cnt; j++) {
        const unsigned long long pos = atomicAdd(count, 1ull);
        if ((long long)pos < cap) { const float4 q = m.pts[(size_t)s * IKD_CELL_PTS + j]; xyz[pos * 3] = q.x; xyz[pos * 3 + 1] = q.y; xyz[pos * 3 + 2] = q.z; }
    }
}

// ---- x = A.colPivHouseholderQr().solve(b), 5 x 3, float (Eigen ColPivHouseholderQR::computeInPlace + _solve_impl, restated) -------------
IMD void ik_qr_solve(float A[5][3], float b[5], float x[3]) {
    const float eps = 1.1920929e-07f;
    float hc[3], nrmU[3], nrmD[3];
    int perm[3] = {0, 1, 2};
    float maxn = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) { float s = 0; for (int r = 0; r < 5; r++) s += A[r][c] * A[r][c]; nrmU[c] = nrmD[c] = sqrtf(s); maxn = fmaxf(maxn, nrmU[c]); }
    const float th0 = maxn * eps / 5.0f, threshold_helper = th0 * th0, downdate = sqrtf(eps);
    int nonzero = 3;
    for (int k = 0; k < 3; k++) {
        int big = k;
        for (int c = k + 1; c < 3; c++) if (nrmU[c] > nrmU[big]) big = c;
        const float bigsq = nrmU[big] * nrmU[big];
        if (nonzero == 3 && bigsq < threshold_helper * (float)(5 - k)) nonzero = k;
        if (big != k) {
            for (int r = 0; r < 5; r++) { const float t = A[r][k]; A[r][k] = A[r][big]; A[r][big] = t; }
            { const float t = nrmU[k]; nrmU[k] = nrmU[big]; nrmU[big] = t; }
            { const float t = nrmD[k]; nrmD[k] = nrmD[big]; nrmD[big] = t; }
            { const int t = perm[k]; perm[k] = perm[big]; perm[big] = t; }
        }
        float tailsq = 0;
        for (int r = k + 1; r < 5; r++) tailsq += A[r][k] * A[r][k];
        const float c0 = A[k][k];
        float beta, tau;
        if (tailsq <= 1.17549435e-38f) { tau = 0; beta = c0; for (int r = k + 1; r < 5; r++) A[r][k] = 0; }
        else {
            beta = sqrtf(c0 * c0 + tailsq);
            if (c0 >= 0) beta = -beta;
            for (int r = k + 1; r < 5; r++) A[r][k] = A[r][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        A[k][k] = beta;
        if (tau != 0)
            for (int c = k + 1; c < 3; c++) {
                float tmp = 0;
                for (int r = k + 1; r < 5; r++) tmp += A[r][k] * A[r][c];
                tmp += A[k][c];
                A[k][c] -= tau * tmp;
                for (int r = k + 1; r < 5; r++) A[r][c] -= tau * A[r][k] * tmp;
            }
        for (int j = k + 1; j < 3; j++)
            if (nrmU[j] != 0) {
                float temp = fabsf(A[k][j]) / nrmU[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0 ? 0 : temp;
                const float q = nrmU[j] / nrmD[j];
                const float temp2 = temp * (q * q);
                if (temp2 <= downdate) {
                    float s = 0;
                    for (int r = k + 1; r < 5; r++) s += A[r][j] * A[r][j];
                    nrmD[j] = sqrtf(s); nrmU[j] = nrmD[j];
                } else nrmU[j] *= sqrtf(temp);
            }
    }
    for (int k = 0; k < nonzero; k++) {
        if (hc[k] == 0) continue;
        float tmp = 0;
        for (int r = k + 1; r < 5; r++) tmp += A[r][k] * b[r];
        tmp += b[k];
        b[k] -= hc[k] * tmp;
        for (int r = k + 1; r < 5; r++) b[r] -= hc[k] * A[r][k] * tmp;
    }
    float c[3] = {0, 0, 0};
    for (int i = nonzero - 1; i >= 0; i--) {
        float s = b[i];
        for (int j = i + 1; j < nonzero; j++) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0;
    for (int i = 0; i < nonzero; i++) x[perm[i]] = c[i];
}
// esti_plane + the gates of voxel_mapping.cpp:1446-1462; returns the new m_point_selected_surf
IMD bool ik_fit_and_gate(const float* nb /* 5 x 3 */, float wx, float wy, float wz, const double* pb, float* normvec) {
    float A[5][3], b[5], nv[3];
#pragma unroll
    for (int j = 0; j < 5; j++) { A[j][0] = nb[j * 3]; A[j][1] = nb[j * 3 + 1]; A[j][2] = nb[j * 3 + 2]; b[j] = -1.0f; }
    ik_qr_solve(A, b, nv);
    const float n = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    const float pa = nv[0] / n, pbn = nv[1] / n, pc = nv[2] / n, pd = (float)(1.0 / (double)n);
    for (int j = 0; j < 5; j++)
        if (fabsf(pa * nb[j * 3] + pbn * nb[j * 3 + 1] + pc * nb[j * 3 + 2] + pd) > 0.05f) return false;
    const float pd2 = pa * wx + pbn * wy + pc * wz + pd;
    const float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2])));
    if (!((double)s > 0.9)) return false;
    normvec[0] = pa; normvec[1] = pbn; normvec[2] = pc; normvec[3] = pd2;
    return true;
}
IMD unsigned long long ik_wave_min(unsigned long long x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long y = __shfl_xor(x, off, 64); x = y < x ? y : x; }
    return x;
}

// One wavefront per query point.  mode 0: plain k-NN (queries are world points); 1: search + fit + gates; 2: fit + gates on the neighbours of
// the last search (nearest_search_en == false, :1423-1443).  The search scans cubes of cells of growing half-width r around the query's cell:
// every map point nearer than r * ds lies inside the cube, so the 5 nearest are final once the 5th distance is below that.
#define IK_WL 1024
__global__ __launch_bounds__(256) void ikd_match_kernel(IkdMapDev m, IkdMatchParams mp, const float* __restrict__ pts, int n, int mode, float* __restrict__ near_xyz,
                                                        int32_t* __restrict__ near_n, int8_t* __restrict__ sel, float* __restrict__ normvec, float* __restrict__ d2_out) {
    __shared__ unsigned long long wl[4][IK_WL];
    __shared__ float nbs[4][IKD_KNN * 3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= n) return;
    const double pb[3] = {(double)pts[(size_t)i * 3], (double)pts[(size_t)i * 3 + 1], (double)pts[(size_t)i * 3 + 2]};
    float wx, wy, wz;
    if (mode == 0) { wx = (float)pb[0]; wy = (float)pb[1]; wz = (float)pb[2]; }
    else {   // pointBodyToWorld (voxel_mapping_common.cpp:121-131): f64 compute, f32 store
        double pi[3], pw[3];
        m3_vec(mp.extR, pb, pi);
        pi[0] += mp.extT[0]; pi[1] += mp.extT[1]; pi[2] += mp.extT[2];
        m3_vec(mp.R, pi, pw);
        wx = (float)(pw[0] + mp.t[0]); wy = (float)(pw[1] + mp.t[1]); wz = (float)(pw[2] + mp.t[2]);
    }
    int nfound = 0;
    bool selected = true;
    if (mode != 2) {
        const long cx = ik_cell(wx, m.ds), cy = ik_cell(wy, m.ds), cz = ik_cell(wz, m.ds);
        int nl = 0;                                   // keys in wl[wv]: (d2 bits << 32) | (slot * 8 + j)
        // matcher: beyond sqrt(5) the point is rejected anyway (pointSearchSqDis[4] > 5, :1436); the plain k-NN hook looks as far as 48 cells
        const int rmax = mode == 0 ? 48 : (int)ceilf(2.2360680f / m.ds) + 1;
        for (int r = 0; r <= rmax; r++) {
            const int w = 2 * r + 1, ncell = w * w * w;
            for (int base = 0; base < ncell; base += 64) {
                const int c = base + lane;
                int cnt = 0;
                long long s = -1;
                if (c < ncell) {
                    const int dz = c % w - r, dy = (c / w) % w - r, dx = c / (w * w) - r;
                    const int ch = max(abs(dx), max(abs(dy), abs(dz)));
                    if (ch == r) { s = ik_find(m, ik_key(cx + dx, cy + dy, cz + dz)); if (s >= 0) cnt = m.count[s]; }
                }
                for (int j = 0; j < IKD_CELL_PTS; j++) {
                    const bool has = j < cnt;
                    if (!__any(has)) break;
                    unsigned long long key = 0;
                    if (has) { const float4 q = m.pts[(size_t)s * IKD_CELL_PTS + j]; key = ((unsigned long long)__float_as_uint(ik_dist(wx, wy, wz, q.x, q.y, q.z)) << 32) | (unsigned long long)((unsigned int)s * 8u + (unsigned int)j); }
                    const unsigned long long mask = __ballot(has);
                    const int pos = nl + __popcll(mask & ((1ull << lane) - 1ull));
                    if (has && pos < IK_WL) wl[wv][pos] = key;
                    nl = min(nl + (int)__popcll(mask), IK_WL);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // keep the 5 smallest keys (ascending) at the head of the list
            const int cnt5 = min(IKD_KNN, nl);
            unsigned long long prev = 0, keep[IKD_KNN];
            for (int k = 0; k < cnt5; k++) {
                unsigned long long mine = ~0ull;
                for (int e = lane; e < nl; e += 64) { const unsigned long long v = wl[wv][e]; if ((k == 0 || v > prev) && v < mine) mine = v; }
                prev = ik_wave_min(mine);
                keep[k] = prev;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) for (int k = 0; k < cnt5; k++) wl[wv][k] = keep[k];
            nl = cnt5;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (nl == IKD_KNN) {
                const float d5 = __uint_as_float((unsigned int)(wl[wv][IKD_KNN - 1] >> 32));
                const float cover = (float)r * m.ds;
                if (d5 <= cover * cover) break;
            }
        }
        nfound = nl;
        if (lane < nfound) {
            const unsigned long long key = wl[wv][lane];
            const unsigned int id = (unsigned int)(key & 0xFFFFFFFFull);
            const float4 q = m.pts[(size_t)(id >> 3) * IKD_CELL_PTS + (id & 7u)];
            nbs[wv][lane * 3] = q.x; nbs[wv][lane * 3 + 1] = q.y; nbs[wv][lane * 3 + 2] = q.z;
            near_xyz[((size_t)i * IKD_KNN + lane) * 3] = q.x; near_xyz[((size_t)i * IKD_KNN + lane) * 3 + 1] = q.y; near_xyz[((size_t)i * IKD_KNN + lane) * 3 + 2] = q.z;
            if (d2_out) d2_out[(size_t)i * IKD_KNN + lane] = __uint_as_float((unsigned int)(key >> 32));
        }
        if (lane == 0) near_n[i] = nfound;
        if (mode == 0) return;
        const float d5 = nfound == IKD_KNN ? __uint_as_float((unsigned int)(wl[wv][IKD_KNN - 1] >> 32)) : 3.4e38f;
        selected = !(d5 > 5.0f);   // m_point_selected_surf[i] = pointSearchSqDis[4] > 5 ? false : true
    } else {
        nfound = near_n[i];
        selected = sel[i] != 0;
        if (lane < IKD_KNN * 3) nbs[wv][lane] = near_xyz[(size_t)i * IKD_KNN * 3 + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;
    bool ok = false;
    if (selected && nfound >= IKD_KNN) {
        float nv4[4];
        ok = ik_fit_and_gate(nbs[wv], wx, wy, wz, pb, nv4);
        if (ok) { normvec[(size_t)i * 4] = nv4[0]; normvec[(size_t)i * 4 + 1] = nv4[1]; normvec[(size_t)i * 4 + 2] = nv4[2]; normvec[(size_t)i * 4 + 3] = nv4[3]; }
    }
    sel[i] = ok ? 1 : 0;
}

// H rows + normal equations (voxel_mapping.cpp:1464-1478, 1487-1590): one thread per point, block sums in lane order, block partials
#define IK_NR 32
__global__ __launch_bounds__(64) void ikd_reduce_kernel(IkdMatchParams mp, const float* __restrict__ body, int n, const int8_t* __restrict__ sel, const float* __restrict__ normvec,
                                                        double* __restrict__ partials) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double acc[IK_NR];
#pragma unroll
    for (int k = 0; k < IK_NR; k++) acc[k] = 0;
    if (i < n && sel[i] && fabs((double)normvec[(size_t)i * 4 + 3]) <= 2.0) {   // m_point_selected_surf[i] && m_res_last[i] <= 2.0
        const double pb[3] = {(double)body[(size_t)i * 3], (double)body[(size_t)i * 3 + 1], (double)body[(size_t)i * 3 + 2]};
        double pt[3], cm[9], T1[9], A[3];
        m3_vec(mp.extR, pb, pt);
        pt[0] += mp.extT[0]; pt[1] += mp.extT[1]; pt[2] += mp.extT[2];
        skew(pt, cm);
        const double nv[3] = {(double)normvec[(size_t)i * 4], (double)normvec[(size_t)i * 4 + 1], (double)normvec[(size_t)i * 4 + 2]};
        m3_mul_bt(cm, mp.R, T1);
        m3_vec(T1, nv, A);
        const double H[6] = {A[0], A[1], A[2], nv[0], nv[1], nv[2]};
        const double meas = -(double)normvec[(size_t)i * 4 + 3];
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double hr = H[r] * mp.r_inv;
#pragma unroll
            for (int c = r; c < 6; c++) acc[k++] = hr * H[c];
            acc[21 + r] = hr * meas;
        }
        acc[27] = 1.0;
        acc[28] = fabs((double)normvec[(size_t)i * 4 + 3]);
    }
    __shared__ double red[IK_NR][65];
#pragma unroll
    for (int k = 0; k < IK_NR; k++) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    if (threadIdx.x < IK_NR) {
        double s = 0;
        for (int j = 0; j < 64; j++) s += red[threadIdx.x][j];
        partials[(size_t)blockIdx.x * IK_NR + threadIdx.x] = s;
    }
}
__global__ __launch_bounds__(64) void ikd_final_kernel(const double* __restrict__ partials, int nb, double* __restrict__ out48) {
    const int k = threadIdx.x;
    __shared__ double tot[IK_NR];
    if (k < IK_NR) { double s = 0; for (int b = 0; b < nb; b++) s += partials[(size_t)b * IK_NR + k]; tot[k] = s; }
    __syncthreads();
    if (k < 48) {
        double v = 0;
        if (k < 36) { const int r = k / 6, c = k % 6; v = tot[r <= c ? ik_sym21(r, c) : ik_sym21(c, r)]; }
        else if (k < 42) v = tot[21 + (k - 36)];
        else if (k < 44) v = tot[27 + (k - 42)];
        out48[k] = v;
    }
}

void launch_ikd_build(hipStream_t s, const IkdMapDev& m, const float* xyz, int n) { KLAUNCH(ikd_build_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, xyz, n); }
void launch_ikd_add(hipStream_t s, const IkdMapDev& m, const float* xyz, int n) {
    KLAUNCH(ikd_add_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, xyz, n);
    KLAUNCH(ikd_add_min_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, xyz, n);
    KLAUNCH(ikd_add_settle_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, xyz);
}
void launch_ikd_delete_boxes(hipStream_t s, const IkdMapDev& m, const IkdBoxes& boxes, int32_t* n_deleted) {
    const long long slots = (long long)m.mask + 1;
    KLAUNCH(ikd_delete_boxes_kernel, dim3((unsigned int)((slots + 255) / 256)), dim3(256), 0, s, m, boxes, n_deleted);
}
void launch_ikd_dump(hipStream_t s, const IkdMapDev& m, float* xyz, long long cap, unsigned long long* count) {
    const long long slots = (long long)m.mask + 1;
    KLAUNCH(ikd_dump_kernel, dim3((unsigned int)((slots + 255) / 256)), dim3(256), 0, s, m, xyz, cap, count);
}
void launch_ikd_match(hipStream_t s, const IkdMapDev& m, const IkdMatchParams& mp, const float* pts, int n, int mode, float* near_xyz, int32_t* near_n, int8_t* sel,
                      float* normvec, float* d2_out) {
    KLAUNCH(ikd_match_kernel, dim3((n + 3) / 4), dim3(256), 0, s, m, mp, pts, n, mode, near_xyz, near_n, sel, normvec, d2_out);
}
void launch_ikd_reduce(hipStream_t s, const IkdMatchParams& mp, const float* body, int n, const int8_t* sel, const float* normvec, double* partials, double* out48) {
    const int nb = (n + 63) / 64;
    KLAUNCH(ikd_reduce_kernel, dim3(nb), dim3(64), 0, s, mp, body, n, sel, normvec, partials);
    KLAUNCH(ikd_final_kernel, dim3(1), dim3(64), 0, s, partials, nb, out48);
}
