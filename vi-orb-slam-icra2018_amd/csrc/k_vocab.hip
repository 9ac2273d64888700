// k_vocab.hip -- SURVEY.md section 8f row 1: the ORB vocabulary tree on the device.
//   * orb_vocab_parse: host parse of the reference's binary vocabulary
//     (ref: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1721 loadFromBinaryFile; layout written by
//     saveToBinaryFile :1727-1751) into structure-of-arrays device tables;
//   * k_vocab_transform: the per-feature transform (ref: TemplatedVocabulary.h:1443-1485, called through
//     Frame::ComputeBoW src/Frame.cc:739-746 with levelsup = 4): descend from the root taking at every
//     level the child with the smallest Hamming distance (FORB::distance, FORB.cpp:82-103; first child
//     wins ties), return the leaf's word id and weight and the node id at level L - levelsup.
// One thread per descriptor: the descriptor stays in 8 VGPRs, each candidate child is two 16-byte
// loads; the upper tree levels are L2-resident, the 32 MB leaf level sits in the Infinity Cache.
// k * L = 60 Hamming distances per feature for the stock vocabulary (k = 10, L = 6).
#include "orbhip_internal.h"

#include <cstring>

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
    // an inner node without children would make the descent loop forever
    for (uint32_t id = 0; id < nb_nodes; id++)
        if (!V.leaf[id] && V.childOff[id] == V.childOff[id + 1]) {
            err = "vocabulary inner node without children";
            return ORBHIP_E_ARG;
        }
    return ORBHIP_OK;
}

__global__ __launch_bounds__(256) void k_vocab_transform(const uint8_t *__restrict__ desc, int n, int nidLevel,
                                                         const uint8_t *__restrict__ vdesc,
                                                         const int32_t *__restrict__ childOff,
                                                         const int32_t *__restrict__ child,
                                                         const uint8_t *__restrict__ leaf,
                                                         const int32_t *__restrict__ word,
                                                         const float *__restrict__ vweight,
                                                         int32_t *__restrict__ word_id, float *__restrict__ weight,
                                                         int32_t *__restrict__ node_id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 a = reinterpret_cast<const uint4 *>(desc + (size_t)i * 32)[0];
    const uint4 b = reinterpret_cast<const uint4 *>(desc + (size_t)i * 32)[1];
    int nid = 0, final_id = 0, level = 0;
    do {
        ++level;
        const int c0 = childOff[final_id], c1 = childOff[final_id + 1];
        int best_d = 1 << 30;
        for (int c = c0; c < c1; c++) {
            const int id = child[c];
            const uint4 p = reinterpret_cast<const uint4 *>(vdesc + (size_t)id * 32)[0];
            const uint4 q = reinterpret_cast<const uint4 *>(vdesc + (size_t)id * 32)[1];
            const int d = __popc(a.x ^ p.x) + __popc(a.y ^ p.y) + __popc(a.z ^ p.z) + __popc(a.w ^ p.w) +
                          __popc(b.x ^ q.x) + __popc(b.y ^ q.y) + __popc(b.z ^ q.z) + __popc(b.w ^ q.w);
            if (d < best_d) {   // strict: the first child wins ties (:1470)
                best_d = d;
                final_id = id;
            }
        }
        if (level == nidLevel) nid = final_id;
    } while (!leaf[final_id]);
    word_id[i] = word[final_id];
    weight[i] = vweight[final_id];
    node_id[i] = nid;
}

void launch_vocab_transform(hipStream_t s, const OrbVocabDev &V, const uint8_t *desc, int n, int levelsup,
                            int32_t *word_id, float *weight, int32_t *node_id)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vocab_transform, dim3((n + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, desc, n, V.L - levelsup,
                       V.desc, V.childOff, V.child, V.leaf, V.word, V.weight, word_id, weight, node_id);
}
