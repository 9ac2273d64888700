// k_vocab.hip -- SURVEY.md section 8f row 1: the ORB vocabulary tree on the device.
//   * orb_vocab_parse: host parse of the reference's binary vocabulary
//     (ref: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1721 loadFromBinaryFile; layout written by
//     saveToBinaryFile :1727-1751) into structure-of-arrays device tables;
//   * k_vocab_transform: the per-feature transform (ref: TemplatedVocabulary.h:1443-1485, called through
//     Frame::ComputeBoW src/Frame.cc:739-746 with levelsup = 4): descend from the root taking at every
//     level the child with the smallest Hamming distance (FORB::distance, FORB.cpp:82-103; first child
//     wins ties), return the leaf's word id and weight and the node id at level L - levelsup.
// One thread per descriptor: the descriptor stays in 8 VGPRs.  The tables are stored by edge (children of a
// node consecutive), so a level is two dependent memory round trips -- the child range of the node just chosen,
// then the descriptors of ten children at a time in flight together -- instead of two per child; the upper
// tree levels are L2-resident, the 32 MB leaf level sits in the Infinity Cache.
// k * L = 60 Hamming distances per feature for the stock vocabulary (k = 10, L = 6).
#include "orbhip_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

int orb_vocab_parse(const uint8_t *blob, size_t nbytes, OrbVocabHost &V, std::string &err)
{
    if (!blob || nbytes < 24) {
        err = "vocabulary blob too small";
        return ORBHIP_E_ARG;
    }
    uint32_t nb_nodes, size_node;
    int32_t hdr[4];
    memcpy(&nb_nodes, blob, 4);
    memcpy(&size_node, blob + 4, 4);
    memcpy(hdr, blob + 8, 16);
    if (size_node != 41) {
        err = "unexpected vocabulary node size (expected 4 + 32 + 4 + 1)";
        return ORBHIP_E_ARG;
    }
    if (nb_nodes < 2 || (nbytes - 24) / 41 != (size_t)nb_nodes - 1) {
        err = "vocabulary blob truncated";
        return ORBHIP_E_ARG;
    }
    V.k = hdr[0];
    V.L = hdr[1];
    V.scoring = hdr[2];
    V.weighting = hdr[3];
    V.nnodes = (int)nb_nodes;
    V.desc.assign((size_t)nb_nodes * 32, 0);
    V.weight.assign(nb_nodes, 0.f);
    V.leaf.assign(nb_nodes, 0);
    V.word.assign(nb_nodes, -1);
    V.childOff.assign((size_t)nb_nodes + 1, 0);
    V.child.assign(nb_nodes, 0);
    std::vector<int32_t> parent(nb_nodes, 0), fill(nb_nodes, 0);
    for (uint32_t id = 1; id < nb_nodes; id++) {
        const uint8_t *r = blob + 24 + (size_t)(id - 1) * 41;
        memcpy(&parent[id], r, 4);
        if (parent[id] < 0 || parent[id] >= (int32_t)nb_nodes) {
            err = "vocabulary node with an out-of-range parent";
            return ORBHIP_E_ARG;
        }
        memcpy(&V.desc[(size_t)id * 32], r + 4, 32);
        memcpy(&V.weight[id], r + 36, 4);
        V.leaf[id] = r[40] ? 1 : 0;
        V.childOff[parent[id] + 1]++;
    }
    for (uint32_t i = 0; i < nb_nodes; i++) V.childOff[i + 1] += V.childOff[i];
    int nwords = 0;
    for (uint32_t id = 1; id < nb_nodes; id++) {
        const int p = parent[id];
        V.child[V.childOff[p] + fill[p]++] = (int32_t)id;   // children keep file order (:1703)
        if (V.leaf[id]) V.word[id] = nwords++;              // words numbered in leaf order (:1707-1712)
    }
    V.nwords = nwords;
    // The reference descends while the node has children (Node::isLeaf() = children.empty(),
    // TemplatedVocabulary.h) and treats the file's is_leaf flag as "this node is a word"; a file in which the two
    // disagree is malformed (an inner node without children would make the descent loop forever).
    for (uint32_t id = 0; id < nb_nodes; id++) {
        const bool noChildren = V.childOff[id] == V.childOff[id + 1];
        if (id > 0 && (V.leaf[id] != 0) != noChildren) {
            err = V.leaf[id] ? "vocabulary leaf node with children" : "vocabulary inner node without children";
            return ORBHIP_E_ARG;
        }
        if (id == 0 && noChildren) {
            err = "vocabulary root without children";
            return ORBHIP_E_ARG;
        }
    }
    const size_t ne = (size_t)nb_nodes - 1;
    V.edesc.resize(ne * 32);
    V.erange.resize(ne * 2);
    V.eword.resize(ne);
    V.eweight.resize(ne);
    for (size_t e = 0; e < ne; e++) {
        const int id = V.child[e];
        memcpy(&V.edesc[e * 32], &V.desc[(size_t)id * 32], 32);
        V.erange[2 * e] = V.childOff[id];
        V.erange[2 * e + 1] = V.childOff[id + 1];
        V.eword[e] = V.word[id];
        V.eweight[e] = V.weight[id];
    }
    return ORBHIP_OK;
}

// ORBVocabulary::loadFromTextFile (ref: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1564-1647; System::System picks it for
// a ".txt" vocabulary, src/System.cc:335-336 -- the stock ORBvoc.txt): first line "k L scoring weighting", then one line per
// node in id order 1, 2, ...: "parent is_leaf d0 ... d31 weight" (the 32 descriptor bytes as decimal integers,
// FORB::fromString; the weight as a decimal double).  The text is converted to the binary layout of saveToBinaryFile
// (:1727-1751) -- which is what tools/bin_vocabulary.cc produces from the same file -- plus the weights as the doubles the
// text loader keeps (Node::weight is a double; the binary file narrows it to float).
// Canonical choice: a line without any token (the empty string getline returns after the file's final newline) is
// skipped.  The reference does not skip it: its `while(!f.eof())` loop runs once more and appends a node whose parent,
// leaf flag and descriptor are whatever the previous iteration and cv::Mat::create left in memory (:1607-1640 with failed
// extractions) -- not a function of the file.
extern "C" int orbhip_vocab_text_to_binary(const char *text, size_t nbytes, void *blob, size_t blob_cap, size_t *blob_bytes,
                                           double *node_weight, size_t weight_cap)
{
    if (!text || !blob_bytes) return ORBHIP_E_ARG;
    const char *p = text, *end = text + nbytes;
    auto next_line = [&](const char *&b, const char *&e) {
        if (p >= end) return false;
        b = p;
        while (p < end && *p != '\n') p++;
        e = p;
        if (p < end) p++;
        return true;
    };
    // tokens of a line: the extraction operators skip white space (std::isspace in the "C" locale)
    auto skip_ws = [](const char *&b, const char *e) {
        while (b < e && (*b == ' ' || *b == '\t' || *b == '\r' || *b == '\v' || *b == '\f')) b++;
    };
    auto get_long = [&](const char *&b, const char *e, long &v) {
        skip_ws(b, e);
        if (b >= e) return false;
        std::string tok(b, std::min<size_t>((size_t)(e - b), 40));
        char *q = nullptr;
        v = strtol(tok.c_str(), &q, 10);
        if (q == tok.c_str()) return false;
        b += q - tok.c_str();
        return true;
    };
    const char *b, *e;
    if (!next_line(b, e)) return ORBHIP_E_ARG;
    long k, L, n1, n2;
    if (!get_long(b, e, k) || !get_long(b, e, L) || !get_long(b, e, n1) || !get_long(b, e, n2)) return ORBHIP_E_ARG;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return ORBHIP_E_ARG;   // :1585-1589
    uint8_t *out = (uint8_t *)blob;
    size_t n = 0;   // nodes written (ids 1..n)
    while (next_line(b, e)) {
        skip_ws(b, e);
        if (b >= e) continue;                                   // canonical: no token on the line
        long pid, leaf;
        if (!get_long(b, e, pid) || !get_long(b, e, leaf)) return ORBHIP_E_ARG;
        if (pid < 0 || pid > (long)n) return ORBHIP_E_ARG;      // m_nodes[pid] must exist (:1614)
        uint8_t d[32];
        for (int i = 0; i < 32; i++) {
            long v;
            if (!get_long(b, e, v)) return ORBHIP_E_ARG;
            d[i] = (uint8_t)v;                                  // FORB::fromString: (unsigned char)n
        }
        skip_ws(b, e);
        if (b >= e) return ORBHIP_E_ARG;
        std::string tok(b, (size_t)(e - b));
        char *q = nullptr;
        const double w = strtod(tok.c_str(), &q);
        if (q == tok.c_str()) return ORBHIP_E_ARG;
        const size_t off = 24 + n * 41;
        if (out && off + 41 <= blob_cap) {
            const int32_t p32 = (int32_t)pid;
            const float wf = (float)w;                          // saveToBinaryFile: _weight = node.weight
            memcpy(out + off, &p32, 4);
            memcpy(out + off + 4, d, 32);
            memcpy(out + off + 36, &wf, 4);
            out[off + 40] = leaf > 0 ? 1 : 0;                   // :1630
        }
        if (node_weight && n < weight_cap) node_weight[n] = w;
        n++;
    }
    *blob_bytes = 24 + n * 41;
    if (out) {
        if (blob_cap < *blob_bytes) return ORBHIP_E_CAPACITY;
        const uint32_t hdr[2] = {(uint32_t)(n + 1), 41u};
        const int32_t h2[4] = {(int32_t)k, (int32_t)L, (int32_t)n1, (int32_t)n2};
        memcpy(out, hdr, 8);
        memcpy(out + 8, h2, 16);
    }
    if (node_weight && weight_cap < n) return ORBHIP_E_CAPACITY;
    return ORBHIP_OK;
}

#define VT_CHUNK 10   // children whose descriptors are in flight together (2 x 16 bytes each; the stock tree has k = 10)

// EAGER_RANGES: the child ranges of ALL candidates travel with their descriptors, so that a level is ONE dependent round trip --
// the chosen child's range is otherwise a second one.  For a frame's worth of descriptors (four workgroups) the descent is
// nothing but its chain of round trips, 12 -> 6 for the stock k = 10, L = 6 tree (r04: orbhip_frame_build 0.163 -> 0.146 ms); a
// batch is bound by the bytes it gathers, and 80 more per level and descriptor cost it 4 % of a step, so batches keep the
// second trip.
template <bool EAGER_RANGES>
__global__ __launch_bounds__(256) void k_vocab_transform(const uint8_t *__restrict__ desc, int n, int nidLevel,
                                                         int rootFirst, int rootLast,
                                                         const uint4 *__restrict__ edesc,
                                                         const int2 *__restrict__ erange,
                                                         const int32_t *__restrict__ eid,
                                                         const int32_t *__restrict__ eword,
                                                         const float *__restrict__ eweight,
                                                         int32_t *__restrict__ word_id, float *__restrict__ weight,
                                                         int32_t *__restrict__ node_id, const int32_t *__restrict__ cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (cnt) n = min(n, cnt[0]);   // (a captured graph: the number of descriptors is the extraction's count, on the device)
    if (i >= n) return;
    const uint4 a = reinterpret_cast<const uint4 *>(desc + (size_t)i * 32)[0];
    const uint4 b = reinterpret_cast<const uint4 *>(desc + (size_t)i * 32)[1];
    int c0 = rootFirst, c1 = rootLast, level = 0, nidEdge = -1, e = 0;
    do {
        ++level;
        // key = distance << 20 | position among the children: the minimum is the first child with the smallest
        // distance (strict '<' over the children in order, :1470)
        unsigned best = 0xFFFFFFFFu;
        int2 br = make_int2(0, 0);
        for (int cb = c0; cb < c1; cb += VT_CHUNK) {
            uint4 p[VT_CHUNK], q[VT_CHUNK];
            int2 rr[VT_CHUNK];
#pragma unroll
            for (int j = 0; j < VT_CHUNK; j++) {          // unconditional loads from clamped edges, issued together
                const int c = min(cb + j, c1 - 1);
                p[j] = edesc[2 * (size_t)c];
                q[j] = edesc[2 * (size_t)c + 1];
                if (EAGER_RANGES) rr[j] = erange[c];
            }
#pragma unroll
            for (int j = 0; j < VT_CHUNK; j++) {
                const unsigned d = __popc(a.x ^ p[j].x) + __popc(a.y ^ p[j].y) + __popc(a.z ^ p[j].z) + __popc(a.w ^ p[j].w) +
                                   __popc(b.x ^ q[j].x) + __popc(b.y ^ q[j].y) + __popc(b.z ^ q[j].z) + __popc(b.w ^ q[j].w);
                const unsigned key = (d << 20) | (unsigned)(cb + j - c0);
                if (cb + j < c1 && key < best) {
                    best = key;
                    if (EAGER_RANGES) br = rr[j];
                }
            }
        }
        e = c0 + (int)(best & 0xFFFFFu);
        if (level == nidLevel) nidEdge = e;
        if (!EAGER_RANGES) br = erange[e];
        c0 = br.x;
        c1 = br.y;
    } while (c0 < c1);
    word_id[i] = eword[e];
    weight[i] = eweight[e];
    node_id[i] = nidEdge >= 0 ? eid[nidEdge] : 0;
}

// Batches: FOUR LANES PER DESCRIPTOR.  The thread-per-descriptor form above requests 20 separate lines per lane and level (two
// 16-byte halves of ten children: every load instruction touches 64 different lines) and is bound by the rate its texture unit
// takes line requests (~1.1 per cycle and CU: 0.18 ms per 1024 frames at 17 % vector issue).  Here a quad takes two children per
// trip -- lane q holds half (q & 1) of child (q >> 1): the quad's four 16-byte loads are 64 consecutive bytes, one request --,
// the halves' popcounts meet by a quad permute, and the smallest key (distance << 20 | position: the first child with the
// smallest distance, the strict '<' over the children in order, :1470) by a second one.  A quarter of the requests for 1.5 x the
// vector instructions.  (r06: two descriptors per quad, their descents interleaved -- twice the loads in flight per wave at 5 .. 7 waves
// per SIMD instead of 8 -- is slower, 0.164 -> 0.167 .. 0.184 ms on random descriptors: the kernel is bound by what the fabric delivers
// (0.56 GB of tree per launch behind the L2s), not by a wave's chain of trips.)
__device__ __forceinline__ unsigned vq_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned vq_swap2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]

#ifndef VT_LPD
#define VT_LPD 4   // lanes per descriptor: 4 (two children per trip), 8 (four) or 16 (eight)
#endif
__global__ __launch_bounds__(256) void k_vocab_transform_quad(const uint8_t *__restrict__ desc, int n, int nidLevel, int rootFirst,
                                                              int rootLast, const uint4 *__restrict__ edesc,
                                                              const int2 *__restrict__ erange, const int32_t *__restrict__ eid,
                                                              const int32_t *__restrict__ eword, const float *__restrict__ eweight,
                                                              int32_t *__restrict__ word_id, float *__restrict__ weight,
                                                              int32_t *__restrict__ node_id)
{
    constexpr int CPT = VT_LPD / 2;                          // children per trip
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = min(t / VT_LPD, n - 1), q = t & (VT_LPD - 1), half = q & 1, sub = q >> 1;   // (a tail group repeats the last descriptor and stores nothing)
    const uint4 a = reinterpret_cast<const uint4 *>(desc + (size_t)i * 32)[half];   // this lane's half of the descriptor
    int c0 = rootFirst, c1 = rootLast, level = 0, nidEdge = -1, e = 0;
    constexpr int TRIPS = (VT_CHUNK + CPT - 1) / CPT;
    do {
        ++level;
        unsigned best = 0xFFFFFFFFu;
        for (int cb = c0; cb < c1; cb += TRIPS * CPT) {
            uint4 p[TRIPS];
#pragma unroll
            for (int j = 0; j < TRIPS; j++) {             // unconditional loads from clamped edges, issued together
                const int c = min(cb + CPT * j + sub, c1 - 1);
                p[j] = edesc[2 * (size_t)c + half];
            }
#pragma unroll
            for (int j = 0; j < TRIPS; j++) {
                unsigned d = __popc(a.x ^ p[j].x) + __popc(a.y ^ p[j].y) + __popc(a.z ^ p[j].z) + __popc(a.w ^ p[j].w);
                d += vq_swap1(d);                          // both halves of the child
                const int c = cb + CPT * j + sub;
                const unsigned key = (d << 20) | (unsigned)(c - c0);
                if (c < c1) best = min(best, key);
            }
        }
        best = min(best, vq_swap2(best));                  // the children columns of a quad, ...
        if (VT_LPD >= 8) best = min(best, (unsigned)__builtin_amdgcn_update_dpp(0, (int)best, 0x141, 0xF, 0xF, true));    // ... of two quads (row_half_mirror), ...
        if (VT_LPD >= 16) best = min(best, (unsigned)__builtin_amdgcn_update_dpp(0, (int)best, 0x140, 0xF, 0xF, true));   // ... of a row (row_mirror)
        e = c0 + (int)(best & 0xFFFFFu);
        if (level == nidLevel) nidEdge = e;
        const int2 br = erange[e];
        c0 = br.x;
        c1 = br.y;
    } while (c0 < c1);
    if (q == 0 && (t / VT_LPD) < n) {
        word_id[i] = eword[e];
        weight[i] = eweight[e];
        node_id[i] = nidEdge >= 0 ? eid[nidEdge] : 0;
    }
}

void launch_vocab_transform(hipStream_t s, const OrbVocabDev &V, const uint8_t *desc, int n, int levelsup,
                            int32_t *word_id, float *weight, int32_t *node_id, const int32_t *cnt)
{
    if (n <= 0) return;
    // A frame's worth of descriptors: workgroups of ONE wave, so that the descent spreads over 16 CUs instead of 4 -- every load of
    // the descent touches 64 different lines, and four waves per CU queue 7 680 line requests per level at its texture unit
    // (r04: the single-frame transform was ~25 us of which ~19 us this queue).  Batches fill the chip either way.
#define ORB_LAUNCH_VT(E, T)                                                                                                       \
    hipLaunchKernelGGL(k_vocab_transform<E>, dim3((n + (T) - 1) / (T), 1, 1), dim3((T), 1, 1), 0, s, desc, n, V.L - levelsup,       \
                       V.rootFirst, V.rootLast, reinterpret_cast<const uint4 *>(V.desc), reinterpret_cast<const int2 *>(V.erange), \
                       V.eid, V.eword, V.eweight, word_id, weight, node_id, cnt)
    static const int quad = ORB_TUNE("VOCAB_QUAD", 1);   // (A/B: 0 = thread per descriptor for batches too)
    if (n <= 16384)
        ORB_LAUNCH_VT(true, 64);
    else if (quad && !cnt)
        hipLaunchKernelGGL(k_vocab_transform_quad, dim3((int)(((size_t)n * VT_LPD + 255) / 256), 1, 1), dim3(256, 1, 1), 0, s, desc, n,
                           V.L - levelsup, V.rootFirst, V.rootLast, reinterpret_cast<const uint4 *>(V.desc),
                           reinterpret_cast<const int2 *>(V.erange), V.eid, V.eword, V.eweight, word_id, weight, node_id);
    else
        ORB_LAUNCH_VT(false, 256);
#undef ORB_LAUNCH_VT
}
