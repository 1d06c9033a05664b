// Context of libvit_amd.so shared by vit_capi.hip (inference) and vit_train.hip (training step).
#pragma once
#include "vit_common.h"

#include <map>
#include <string>
#include <vector>

struct Slot {
    size_t offset;      // bytes into the weight arena
    int64_t rows, cols; // source shape as a matrix (vectors: rows = 1)
    int64_t pitch;      // destination row pitch in elements
    int dtype;          // storage dtype in the arena
    bool loaded;
    size_t goff;        // element offset of the same tensor in the fp32 gradient / master arenas (same pitch)
};

struct LayerW {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv, *bo, *b1, *b2;
    void *wqkv, *wo, *w1, *w2;
};

// LayerNorm folded into the GEMM that follows it (inference forward, bf16): per layer the gamma-scaled
// copies of the QKV / fc1 weights, their column sums c and the folded biases d (ln_fold_kernel, layernorm.hip)
struct LnFold {
    void *wqkv, *w1;
    float *cqkv, *dqkv, *c1, *d1;
};

// fp32-arena element offsets of one layer's tensors (gradients and master weights share them)
struct LayerG {
    size_t ln1_g, ln1_b, ln2_g, ln2_b, bqkv, bo, b1, b2, wqkv, wo, w1, w2;
    size_t begin, end;  // [begin, end): the layer's contiguous range (one all-reduce bucket)
};

struct TrainState;  // vit_train.hip

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t esize(int dt) { return dt == VIT_F32 ? 4 : 2; }

struct vit_ctx {
    vit_config cfg;
    int device;
    int N, dh, kreal, kpad;
    unsigned char* arena;
    size_t arena_bytes;
    std::map<std::string, Slot> slots;
    std::vector<std::string> optional;  // names that may legitimately stay unloaded (none today)
    std::string missing_name;
    int n_missing;
    // resolved pointers
    void* patch_w;
    float *patch_b, *cls, *pos, *pre_g, *pre_b, *fin_g, *fin_b;
    std::vector<LayerW> layers;
    // fp32 gradient-arena layout
    size_t g_total;  // floats
    size_t g_patch_w, g_patch_b, g_cls, g_pos, g_pre_g, g_pre_b, g_fin_g, g_fin_b;
    std::vector<LayerG> glayers;
    TrainState* train;
    // two-stream forward (vit_capi.hip): the two chunks of a batch run on the caller's stream and on
    // aux_stream, so one half's kernels fill the CUs the other half's last tile round leaves idle
    static const int kMaxStreams = 4;
    hipStream_t aux_stream[kMaxStreams - 1];
    hipEvent_t ev_fork, ev_join[kMaxStreams - 1];
    int nstreams;  // 1 = single stream
    // folded LayerNorm (vit_capi.hip): copies are refreshed on the forward's stream whenever a weight changed
    int ln_fuse;               // 0 = off (fp32 contexts, VIT_AMD_FUSE_LN=0)
    // patch sizes the persistent GEMM's pixel loader does not take (P = 14): the forward gathers the patch rows
    // into a dense [B * np, patch_kp] matrix and multiplies by a copy of the patch weight padded to patch_kp
    int patch_kp;              // 0 = route not used; else C*P*P rounded up to a multiple of 128
    void* patch_w_kp;          // [D, patch_kp] bf16, zero padded (in fold_arena, refreshed with the folds)
    unsigned char* fold_arena;
    std::vector<LnFold> folds;
    bool fold_dirty;
};
