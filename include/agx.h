/*
 * agx.h — C ABI of the MI355X-native self-play MCTS + policy/value evaluation engine.
 *
 * This is the drop-in boundary for the hot path of AlphaGomoku's src/search + src/selfplay +
 * src/networks. The reference has no FFI layer on this path (its boundary is a set of C++ classes over
 * the MinML C++ API), so the entry points below are what a cgo/ctypes/C++ facade for that path binds.
 * The only extern "C" surface in the reference is the dataset reader
 * include/alphagomoku/dataset/torch_api.h:13-43; its conventions are kept: plain structs, caller-owned
 * flat buffers, sizes queried first.  Unlike torch_api.h every call returns an int status (0 = ok) and
 * agx_last_error() returns the message — the C++ facade turns a non-zero status into the same
 * std::logic_error / std::runtime_error the reference throws (NNEvaluator.cpp:149,185-187).
 *
 * No torch types, no C++ types, no HIP types appear in the signatures.  Pointers named d_* are device
 * (HBM) addresses, h_* are host addresses, `stream` is a hipStream_t passed as void* (NULL = default).
 *
 * One handle per GPU, each driven by exactly one host thread (mirrors GeneratorThread,
 * src/selfplay/GeneratorManager.cpp:124-141).
 */
#ifndef AGX_H_
#define AGX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_OK 0
#define AGX_ERR_INVALID 1
#define AGX_ERR_HIP 2
#define AGX_ERR_UNSUPPORTED 3
#define AGX_ERR_STATE 4

/* Game rules, same numbering as ag::GameRules (include/alphagomoku/game/rules.hpp:18-25). */
enum AgxRules { AGX_FREESTYLE = 0, AGX_STANDARD = 1, AGX_RENJU = 2, AGX_CARO5 = 3, AGX_CARO6 = 4 };

/* Signs, same numbering as ag::Sign (include/alphagomoku/game/Move.hpp:17-23). */
enum AgxSign { AGX_NONE = 0, AGX_CROSS = 1, AGX_CIRCLE = 2, AGX_ILLEGAL = 3 };

const char* agx_last_error(void);
int agx_version(void);
/* Selects the HIP device for the calling thread (one engine per GPU). */
int agx_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * Policy/value network (replaces AGNetwork::forward / asyncForwardLaunch+Join over ml::Graph,
 * src/networks/AGNetwork.cpp:61-88, for the ResnetPV architecture of src/networks/networks.cpp:71-93
 * built from src/networks/blocks.cpp:32-38,45-55,99-118 after optimize(2) has folded the BNs).
 * ---------------------------------------------------------------------------------------------- */
typedef struct AgxNetDesc
{
	int rows;           /* board rows (15 or 20) */
	int cols;           /* board cols */
	int blocks;         /* residual blocks */
	int filters;        /* conv filters F (64 or 128) */
	int in_channels;    /* 32 (bit-packed features, NNInputFeatures.cpp:59-113) */
	int value_hidden;   /* D = min(256, 2F) (blocks.cpp:113) */
} AgxNetDesc;

typedef struct AgxNet AgxNet; /* opaque */

/* Number of fp32 values in the canonical (BN-folded) weight blob, in this order:
 *   conv_in  W[5][5][Cin][F]  b[F]
 *   blocks x { W1[3][3][F][F] b1[F]  W2[3][3][F][F] b2[F] }
 *   policy   Wp1[3][3][F][F] bp1[F]  Wp2[F] bp2[1]
 *   value    Wv1[F][4] bv1[4]  Wv2[rows*cols*4][D] bv2[D]  Wv3[D][3] bv3[3]
 * Conv weights are [kh][kw][cin][cout] (cross-correlation, "same" zero padding); the value-head flatten
 * order is NHWC row-major: index = (row*cols + col)*4 + c.  */
size_t agx_net_blob_floats(const AgxNetDesc* desc);
int agx_net_create(const AgxNetDesc* desc, AgxNet** out);
int agx_net_load_weights(AgxNet* net, const float* h_blob, size_t n_floats);
/* d_features: uint32[batch][rows*cols] (one bit-packed word per cell);
 * d_policy: float[batch][rows*cols] (softmax over the board); d_value: float[batch][3] = (win, draw, loss). */
int agx_nn_forward(AgxNet* net, const uint32_t* d_features, int batch, float* d_policy, float* d_value, void* stream);
int agx_net_destroy(AgxNet* net);

/* Raw device-memory helpers so that non-HIP hosts (ctypes, cgo) can stage buffers. */
int agx_malloc(void** d_ptr, size_t bytes);
int agx_free(void* d_ptr);
int agx_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int agx_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int agx_memset(void* d_ptr, int value, size_t bytes);
int agx_device_synchronize(void);

/* Timing of work enqueued on `stream` with HIP events recorded on that same stream. */
typedef struct AgxTimer AgxTimer;
int agx_timer_create(AgxTimer** out);
int agx_timer_start(AgxTimer* t, void* stream);
int agx_timer_stop(AgxTimer* t, void* stream);
int agx_timer_elapsed_ms(AgxTimer* t, float* ms); /* synchronises on the stop event */
int agx_timer_destroy(AgxTimer* t);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H_ */
