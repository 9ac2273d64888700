#!/bin/bash
# Experiment build: liborbhip with per-barrier time stamps in the quadtree workgroups (build_ab/lib_qtstamp.so);
# read with tools/scratch/qt_stamps.py on the GPU box after copying it over csrc/liborbhip.so.
set -e
cd /root/repo
rm -rf build_ab/csrc_stamp && mkdir -p build_ab && cp -r vi-orb-slam-icra2018_amd/csrc build_ab/csrc_stamp
cd build_ab/csrc_stamp && rm -f *.o *.so
sed -i 's/x\.sync()/x.sync_id(__LINE__)/g' quadtree_core.h
python3 - <<'PY2'
p='quadtree_core.h'
s=open(p).read()
s=s.replace('                    r = x.group_sum(r, sl);   // called by every thread of the workgroup','                    x.sync_id(9001);\n                    r = x.group_sum(r, sl);   // called by every thread of the workgroup\n                    x.sync_id(9002);')
s=s.replace('                const int SPLIT = 1 << sl;\n                for (int i0 = 0;','                const int SPLIT = 1 << sl;\n                x.sync_id(9000);\n                for (int i0 = 0;')
open(p,'w').write(s)
PY2
python3 - <<'PY'
p='k_quadtree.hip'
s=open(p).read()
s=s.replace('''template <bool LAT>
struct QtBlock {''','''__device__ unsigned long long g_qt_stamps[16][512];
__device__ int g_qt_nstamps[16];
template <bool LAT>
struct QtBlock {
    int lvl;
    mutable int ns;
    __device__ __forceinline__ void sync_id(int line) const
    {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x == 0 && ns < 512) g_qt_stamps[lvl][ns++] = ((unsigned long long)line << 40) | (wall_clock64() & 0xFFFFFFFFFFull);
    }''')
s=s.replace('''    x.wtot = s_wtot;''','''    x.wtot = s_wtot;
    x.lvl = l;
    x.ns = 0;
    if (tid == 0 && blockIdx.x == 0) g_qt_stamps[l][x.ns++] = wall_clock64() & 0xFFFFFFFFFFull;''')
s=s.replace('''        if (tid == 0) lvlKpCnt[frame * ORBHIP_MAX_LEVELS + l] = S;
    };''','''        if (tid == 0) lvlKpCnt[frame * ORBHIP_MAX_LEVELS + l] = S;
        if (tid == 0 && blockIdx.x == 0) {
            g_qt_stamps[l][x.ns++] = wall_clock64() & 0xFFFFFFFFFFull;
            g_qt_nstamps[l] = x.ns;
        }
    };''')
s=s.replace('''size_t quadtree_lds_bytes(const OrbLevels &G)''','''extern "C" int orbhip_debug_qt_stamps(unsigned long long *stamps, int *n)
{
    if (hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_qt_stamps), sizeof(unsigned long long) * 16 * 512) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_qt_nstamps), sizeof(int) * 16) != hipSuccess) return -1;
    return 0;
}

size_t quadtree_lds_bytes(const OrbLevels &G)''')
assert 'orbhip_debug_qt_stamps' in s
open(p,'w').write(s)
PY
make 2>&1 | grep -E "error" || true
cp liborbhip.so ../lib_qtstamp.so
