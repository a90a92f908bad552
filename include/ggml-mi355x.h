/*
 * ggml-mi355x.h — public C API of the MI355X ggml backend (the counterpart of the reference's
 * ggml/include/ggml-hexagon.h:31-50, same shape: init / is_<name> / device count / reg / device name).
 *
 * The backend registers through ggml's plugin ABI (ggml/src/ggml-backend-impl.h) exactly as
 * ggml_backend_hexagon_reg does (ggml/src/ggml-hexagon/ggml-hexagon.cpp:5941-6007, GGML_BACKEND_DL_IMPL :6127):
 * built as a GGML_BACKEND_DL module it exports ggml_backend_init / ggml_backend_score and is picked up by
 * unmodified llama.cpp binaries through GGML_BACKEND_PATH (ggml/src/ggml-backend-reg.cpp:589-593).
 * Offloaded surface: GGML_OP_MUL_MAT and GGML_OP_MUL_MAT_ID with src0 in {Q4_0, Q8_0, Q4_K, Q5_K, Q6_K},
 * src1/dst F32 — the quantized mulmat the reference's cDSP path gates in
 * ggmlhexagon_can_handle_op_through_cdsp (ggml-hexagon.cpp:5065-5115) — plus, so that a transformer layer is one scheduler
 * split and the KV cache lives in HBM, the ops around it (include/ggml_mi355x_ops.h): ADD SUB MUL DIV SCALE,
 * SILU GELU RELU TANH SIGMOID NEG EXP, NORM, RMS_NORM, ROPE, SOFT_MAX, CPY CONT DUP, GET_ROWS, ARGSORT, SUM_ROWS and MUL_MAT with an
 * F16 / F32 src0.  GGML_MI355X_GLUE=0 restricts the device to the two quantized ops.
 */
#pragma once

#include "ggml.h"
#include "ggml-backend.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_MAX_DEVICES   8
#define GGML_MI355X_BACKEND_NAME  "MI355X"

/* dev_num: HIP device ordinal (only gfx950 devices are accepted).  Returns NULL on failure. */
GGML_BACKEND_API ggml_backend_t     ggml_backend_mi355x_init(size_t dev_num);

GGML_BACKEND_API bool               ggml_backend_is_mi355x(ggml_backend_t backend);

GGML_BACKEND_API int                ggml_backend_mi355x_get_device_count(void);

GGML_BACKEND_API ggml_backend_reg_t ggml_backend_mi355x_reg(void);

GGML_BACKEND_API const char *       ggml_backend_mi355x_get_devname(size_t dev_num);

/* device buffer type of a device (weights / activations in HBM, is_host = false) */
GGML_BACKEND_API ggml_backend_buffer_type_t ggml_backend_mi355x_buffer_type(size_t dev_num);

#ifdef __cplusplus
}
#endif
