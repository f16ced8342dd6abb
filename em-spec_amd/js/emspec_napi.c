/*
 * emspec_napi.c — thin N-API (raw node_api.h, N-API >= 4) shim over the C ABI of
 * libemspec (include/emspec.h).  This is the process-internal FFI boundary of
 * renderer JS -> this shim -> libemspec -> HIP kernels.
 *
 * The reference ships no addon, binding.gyp or IPC schema (its source is private,
 * /root/reference/README.md:73); the only interface it names is the renderer call
 * computeSpectrogramColumn(audioFrame, fftSize, hop, reassign) (BASELINE.json).
 *
 * Ownership: typed-array memory is borrowed for the duration of a call only.
 * Errors: every failure throws a JS Error whose .code is the emspec_status name.
 * There is no CPU path: without a gfx950 device create() throws EMSPEC_ERR_NO_DEVICE.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/emspec.h"

#define NAPI_OK_OR_RETURN(env, call)                                     \
    do {                                                                 \
        if ((call) != napi_ok) {                                         \
            napi_throw_error((env), "EMSPEC_NAPI", "N-API call failed: " #call); \
            return NULL;                                                 \
        }                                                                \
    } while (0)

static const char* status_name(int rc) {
    switch (rc) {
        case EMSPEC_ERR_INVALID_ARG: return "EMSPEC_ERR_INVALID_ARG";
        case EMSPEC_ERR_NO_DEVICE: return "EMSPEC_ERR_NO_DEVICE";
        case EMSPEC_ERR_HIP: return "EMSPEC_ERR_HIP";
        case EMSPEC_ERR_OUT_OF_MEMORY: return "EMSPEC_ERR_OUT_OF_MEMORY";
        case EMSPEC_ERR_STATE: return "EMSPEC_ERR_STATE";
        case EMSPEC_ERR_COMM: return "EMSPEC_ERR_COMM";
        default: return "EMSPEC_ERR_UNKNOWN";
    }
}

static napi_value throw_status(napi_env env, emspec_engine* e, int rc) {
    napi_throw_error(env, status_name(rc), emspec_last_error(e));
    return NULL;
}

typedef struct { emspec_engine* e; int32_t rows; } handle_t;   /* rows: the engine's row count, for output-size checks */

static void finalize_handle(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    handle_t* h = (handle_t*)data;
    if (h) { if (h->e) emspec_destroy(h->e); free(h); }
}

static int get_number_prop(napi_env env, napi_value obj, const char* name, double* out) {
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
    napi_value v;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return 0;
    return napi_get_value_double(env, v, out) == napi_ok;
}

static handle_t* get_handle(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((handle_t*)p)->e) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "not a live emspec engine handle");
        return NULL;
    }
    return (handle_t*)p;
}

/* typed array of the wanted element type, or NULL data when the value is undefined/null */
static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len, int optional) {
    *data = NULL; *len = 0;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (optional && (t == napi_undefined || t == napi_null)) return 1;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return 0;
    napi_typedarray_type ty; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, &ty, len, data, &ab, &off) != napi_ok) return 0;
    return ty == want;
}

/* create(config) -> external */
static napi_value Create(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    emspec_config cfg;
    emspec_default_config(&cfg);
    if (argc >= 1) {
        napi_valuetype t;
        NAPI_OK_OR_RETURN(env, napi_typeof(env, argv[0], &t));
        if (t == napi_object) {
            double d;
            if (get_number_prop(env, argv[0], "device", &d)) cfg.device = (int32_t)d;
            if (get_number_prop(env, argv[0], "rows", &d)) cfg.rows = (int32_t)d;
            if (get_number_prop(env, argv[0], "sampleRate", &d)) cfg.sample_rate = (float)d;
            if (get_number_prop(env, argv[0], "fminHz", &d)) cfg.fmin_hz = (float)d;
            if (get_number_prop(env, argv[0], "fmaxHz", &d)) cfg.fmax_hz = (float)d;
            if (get_number_prop(env, argv[0], "gain", &d)) cfg.gain = (float)d;
            if (get_number_prop(env, argv[0], "dbTop", &d)) cfg.db_top = (float)d;
            if (get_number_prop(env, argv[0], "dbRange", &d)) cfg.db_range = (float)d;
            if (get_number_prop(env, argv[0], "gateDb", &d)) cfg.gate_db = (float)d;
            if (get_number_prop(env, argv[0], "powerFloor", &d)) cfg.power_floor = (float)d;
            /* exact: true -> EMSPEC_MODE_EXACT (binary64 + 64-bit fixed-point histogram: indices equal to a float64
             * implementation, reproducible bytes); default EMSPEC_MODE_FAST */
            napi_value ex; bool has = false, on = false;
            if (napi_has_named_property(env, argv[0], "exact", &has) == napi_ok && has &&
                napi_get_named_property(env, argv[0], "exact", &ex) == napi_ok &&
                napi_coerce_to_bool(env, ex, &ex) == napi_ok && napi_get_value_bool(env, ex, &on) == napi_ok && on)
                cfg.mode = EMSPEC_MODE_EXACT;
        }
    }
    emspec_engine* e = NULL;
    int rc = emspec_create(&cfg, &e);
    if (rc != EMSPEC_OK) return throw_status(env, NULL, rc);
    if (emspec_mode(e) != cfg.mode) {   /* a library that ignores the mode field must not pass for an exact engine */
        emspec_destroy(e);
        napi_throw_error(env, "EMSPEC_ERR_STATE", "the library did not honour the requested arithmetic mode");
        return NULL;
    }
    handle_t* h = (handle_t*)malloc(sizeof(handle_t));
    if (!h) { emspec_destroy(e); napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "malloc"); return NULL; }
    h->e = e;
    h->rows = cfg.rows;
    napi_value ext;
    if (napi_create_external(env, h, finalize_handle, NULL, &ext) != napi_ok) {
        finalize_handle(env, h, NULL);
        napi_throw_error(env, "EMSPEC_NAPI", "napi_create_external failed");
        return NULL;
    }
    return ext;
}

static napi_value Destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void* p = NULL;
    if (argc >= 1 && napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        handle_t* h = (handle_t*)p;
        if (h->e) { emspec_destroy(h->e); h->e = NULL; }
    }
    return NULL;
}

static napi_value Rows(napi_env env, napi_callback_info info) {
    /* rows(config) -> rows the engine would use (defaults applied) */
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    emspec_config cfg; emspec_default_config(&cfg);
    double d;
    if (argc >= 1 && get_number_prop(env, argv[0], "rows", &d)) cfg.rows = (int32_t)d;
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int32(env, cfg.rows, &r));
    return r;
}

/* column(handle, frame:Float32Array, fftSize, hop, reassign, outDb?:Float32Array, outRgba?:Uint8Array) -> column index (-1 while priming) */
static napi_value Column(napi_env env, napi_callback_info info) {
    size_t argc = 7; napi_value argv[7];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 5) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "column(handle, frame, fftSize, hop, reassign[, outDb, outRgba])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* frame; size_t flen;
    if (!get_typed(env, argv[1], napi_float32_array, &frame, &flen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "audioFrame must be a Float32Array"); return NULL; }
    int32_t n, hop; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[4], &argv[4]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[4], &reassign));
    if ((size_t)n != flen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "audioFrame.length must equal fftSize"); return NULL; }
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0;
    if (argc > 5 && !get_typed(env, argv[5], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 6 && !get_typed(env, argv[6], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    int32_t rows = db ? (int32_t)dblen : (int32_t)(rgbalen / 4);
    if (rgba && db && rgbalen != 4 * dblen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba.length must be 4*outDb.length"); return NULL; }
    int64_t col = -1;
    int rc = emspec_column(h->e, (const float*)frame, n, hop, reassign ? 1 : 0, (float*)db, (uint8_t*)rgba, rows, &col);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, col, &r));
    return r;
}

/* flush(handle, outDb?, outRgba?) -> column index */
static napi_value Flush(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 2) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "flush(handle, outDb[, outRgba])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0;
    if (!get_typed(env, argv[1], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 2 && !get_typed(env, argv[2], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    int32_t rows = db ? (int32_t)dblen : (int32_t)(rgbalen / 4);
    int64_t col = -1;
    int rc = emspec_column_flush(h->e, (float*)db, (uint8_t*)rgba, rows, &col);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, col, &r));
    return r;
}

/* pushColumns(handle, count, fftSize, hop, reassign) -> columns a block of `count` samples will complete */
static napi_value PushColumns(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 5) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pushColumns(handle, count, fftSize, hop, reassign)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    int64_t count; int32_t n, hop; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[1], &count));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[4], &argv[4]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[4], &reassign));
    const int64_t k = emspec_push_columns(h->e, count, n, hop, reassign ? 1 : 0);
    if (k < 0) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "invalid count / fftSize / hop"); return NULL; }
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, k, &r));
    return r;
}

/* push(handle, samples:Float32Array, fftSize, hop, reassign, rows, outDb?:Float32Array(k*rows), outRgba?:Uint8Array(4*k*rows))
 * -> absolute index of the first completed column (-1 when the block completed none) */
static napi_value Push(napi_env env, napi_callback_info info) {
    size_t argc = 8; napi_value argv[8];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 6) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "push(handle, samples, fftSize, hop, reassign, rows[, outDb, outRgba])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* smp; size_t slen;
    if (!get_typed(env, argv[1], napi_float32_array, &smp, &slen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "samples must be a Float32Array"); return NULL; }
    int32_t n, hop, rows; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[4], &argv[4]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[4], &reassign));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &rows));
    if (rows < 1) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "rows must be positive"); return NULL; }
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0;
    if (argc > 6 && !get_typed(env, argv[6], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 7 && !get_typed(env, argv[7], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    int64_t room = INT64_MAX;   /* columns the smaller output can hold */
    if (db) room = (int64_t)(dblen / (size_t)rows);
    if (rgba && (int64_t)(rgbalen / (4 * (size_t)rows)) < room) room = (int64_t)(rgbalen / (4 * (size_t)rows));
    int64_t count = 0, first = -1;
    int rc = emspec_push_samples(h->e, (const float*)smp, (int64_t)slen, n, hop, reassign ? 1 : 0, (float*)db, (uint8_t*)rgba,
                                 rows, room, &count, &first);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, first, &r));
    return r;
}

/* warpedEdges(rows, fminHz, fmaxHz, lowEndBoost, freqScale) -> Float32Array(rows+1)  (emspec_warped_edges_hz) */
static napi_value WarpedEdges(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 5) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "warpedEdges(rows, fminHz, fmaxHz, lowEndBoost, freqScale)"); return NULL; }
    int32_t rows; double a[4];
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[0], &rows));
    for (int i = 0; i < 4; ++i) NAPI_OK_OR_RETURN(env, napi_get_value_double(env, argv[1 + i], &a[i]));
    if (rows < 1 || rows > (1 << 20)) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "rows out of range"); return NULL; }
    napi_value ab, ta; void* data = NULL;
    NAPI_OK_OR_RETURN(env, napi_create_arraybuffer(env, (size_t)(rows + 1) * 4, &data, &ab));
    if (emspec_warped_edges_hz(rows, (float)a[0], (float)a[1], (float)a[2], (float)a[3], (float*)data) != EMSPEC_OK) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "need 0 < fminHz < fmaxHz, lowEndBoost > 0, freqScale > 0");
        return NULL;
    }
    NAPI_OK_OR_RETURN(env, napi_create_typedarray(env, napi_float32_array, (size_t)rows + 1, ab, 0, &ta));
    return ta;
}

/* referenceColormap(brightness) -> Uint8Array(1024)  (emspec_make_colormap) */
static napi_value ReferenceColormap(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    double b = 0.5;
    if (argc >= 1) NAPI_OK_OR_RETURN(env, napi_get_value_double(env, argv[0], &b));
    napi_value ab, ta; void* data = NULL;
    NAPI_OK_OR_RETURN(env, napi_create_arraybuffer(env, 1024, &data, &ab));
    if (emspec_make_colormap((float)b, (uint8_t*)data) != EMSPEC_OK) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "brightness must be >= 0"); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_create_typedarray(env, napi_uint8_array, 1024, ab, 0, &ta));
    return ta;
}

static napi_value Reset(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    int rc = emspec_reset(h->e);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

/* batch(handle, pcm:Float32Array(S*L), S, L, fftSize, hop, reassign, outDb?, outRgba?, outIndex?) -> columns per stream */
static napi_value Batch(napi_env env, napi_callback_info info) {
    size_t argc = 10; napi_value argv[10];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 8) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "batch(handle, pcm, S, L, fftSize, hop, reassign, outDb[, outRgba, outIndex])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* pcm; size_t plen;
    if (!get_typed(env, argv[1], napi_float32_array, &pcm, &plen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm must be a Float32Array"); return NULL; }
    int32_t S, n, hop; int64_t L; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[3], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[6], &argv[6]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[6], &reassign));
    if (S < 1 || L < 1 || (size_t)S * (size_t)L != plen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm.length must equal S*L"); return NULL; }
    emspec_out out; memset(&out, 0, sizeof(out));
    size_t l0 = 0, l1 = 0, l2 = 0; void *p0 = NULL, *p1 = NULL, *p2 = NULL;
    if (!get_typed(env, argv[7], napi_float32_array, &p0, &l0, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 8 && !get_typed(env, argv[8], napi_uint8_array, &p1, &l1, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 9 && !get_typed(env, argv[9], napi_uint8_array, &p2, &l2, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outIndex must be a Uint8Array"); return NULL; }
    int64_t C = emspec_num_columns(L, n, hop);
    /* sizes are checked against rows implied by the first output given */
    size_t cells = 0;
    if (p0) cells = l0; else if (p1) cells = l1 / 4; else if (p2) cells = l2;
    /* emspec_batch writes S*columns*rows cells with the ENGINE's row count: anything else would overrun the array */
    if (C <= 0 || cells != (size_t)S * (size_t)C * (size_t)h->rows || (p1 && l1 != 4 * cells) || (p2 && l2 != cells) || (p0 && l0 != cells)) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "output arrays must hold exactly S*columns*rows cells, rows = the engine's row count (rgba: 4 bytes per cell)");
        return NULL;
    }
    out.db = (float*)p0; out.rgba = (uint8_t*)p1; out.index = (uint8_t*)p2;
    int rc = emspec_batch(h->e, (const float*)pcm, S, L, n, hop, reassign ? 1 : 0, &out);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, C, &r));
    return r;
}

/* batchPacked(handle, pcm:Float32Array(S*L), S, L, fftSize, hop, reassign, wire:Uint8Array, offsets:Float64Array(S+1)) -> columns
 * per stream.  emspec_batch_packed: the palette-index columns cross PCIe as one lossless wire image per stream; stream s
 * is wire.subarray(offsets[s], offsets[s+1]) (offsets as doubles: exact below 2^53). */
static napi_value BatchPacked(napi_env env, napi_callback_info info) {
    size_t argc = 9; napi_value argv[9];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 9) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "batchPacked(handle, pcm, S, L, fftSize, hop, reassign, wire, offsets)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* pcm; size_t plen;
    if (!get_typed(env, argv[1], napi_float32_array, &pcm, &plen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm must be a Float32Array"); return NULL; }
    int32_t S, n, hop; int64_t L; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[3], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[6], &argv[6]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[6], &reassign));
    if (S < 1 || L < 1 || (size_t)S * (size_t)L != plen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm.length must equal S*L"); return NULL; }
    void *wire = NULL, *offs = NULL; size_t wlen = 0, olen = 0;
    if (!get_typed(env, argv[7], napi_uint8_array, &wire, &wlen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "wire must be a Uint8Array"); return NULL; }
    if (!get_typed(env, argv[8], napi_float64_array, &offs, &olen, 0) || olen != (size_t)S + 1) {
        napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "offsets must be a Float64Array(S + 1)"); return NULL;
    }
    int64_t* o64 = (int64_t*)malloc(sizeof(int64_t) * ((size_t)S + 1));
    if (!o64) { napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "out of host memory"); return NULL; }
    int rc = emspec_batch_packed(h->e, (const float*)pcm, S, L, n, hop, reassign ? 1 : 0, (uint8_t*)wire, (int64_t)wlen, o64);
    if (rc == EMSPEC_OK) for (int32_t i = 0; i <= S; ++i) ((double*)offs)[i] = (double)o64[i];
    free(o64);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, emspec_num_columns(L, n, hop), &r));
    return r;
}

/* wireUnpack(image:Uint8Array, columns, rows, out:Uint8Array(columns*rows)): emspec_wire_unpack_host (no device, no engine) */
static napi_value WireUnpack(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 4) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "wireUnpack(image, columns, rows, out)"); return NULL; }
    void *img = NULL, *out = NULL; size_t ilen = 0, olen = 0;
    int64_t columns; int32_t rows;
    if (!get_typed(env, argv[0], napi_uint8_array, &img, &ilen, 0) || !get_typed(env, argv[3], napi_uint8_array, &out, &olen, 0)) {
        napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "image and out must be Uint8Arrays"); return NULL;
    }
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[1], &columns));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &rows));
    if (columns < 1 || rows < 1 || olen != (size_t)columns * (size_t)rows) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "out.length must equal columns*rows"); return NULL; }
    if (emspec_wire_unpack_host((const uint8_t*)img, (int64_t)ilen, columns, rows, (uint8_t*)out) != EMSPEC_OK) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "the wire image does not match (columns, rows) or is damaged"); return NULL;
    }
    return NULL;
}

static napi_value WireBound(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int64_t columns = 0; int32_t rows = 0;
    if (argc < 2) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "wireBound(columns, rows)"); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[0], &columns));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[1], &rows));
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, emspec_wire_bound(columns, rows), &r));
    return r;
}

/* batchAsync(handle, pcm, S, L, fftSize, hop, reassign, outDb?, outRgba?, outIndex?) -> Promise<columns>
 * Same as batch() but the work runs on the libuv thread pool (napi_create_async_work), so the
 * renderer's JS thread is not blocked for the duration of a large batch.  The typed arrays are
 * kept alive by references until completion; the caller must not touch them, nor call into the
 * same engine, before the promise settles (an engine is not thread-safe). */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[4];
    emspec_engine* e;
    const float* pcm;
    emspec_out out;
    int32_t S, n, hop, reassign;
    int64_t L, C;
    int rc;
    char msg[256];
} batch_job;

static void batch_execute(napi_env env, void* data) {
    (void)env;
    batch_job* j = (batch_job*)data;
    j->rc = emspec_batch(j->e, j->pcm, j->S, j->L, j->n, j->hop, j->reassign, &j->out);
    if (j->rc != EMSPEC_OK) { strncpy(j->msg, emspec_last_error(j->e), sizeof(j->msg) - 1); j->msg[sizeof(j->msg) - 1] = 0; }
}

static void batch_complete(napi_env env, napi_status status, void* data) {
    batch_job* j = (batch_job*)data;
    for (int i = 0; i < 4; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
    if (status == napi_ok && j->rc == EMSPEC_OK) {
        napi_value v; napi_create_int64(env, j->C, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value code, msg, err;
        napi_create_string_utf8(env, status == napi_ok ? status_name(j->rc) : "EMSPEC_NAPI", NAPI_AUTO_LENGTH, &code);
        napi_create_string_utf8(env, status == napi_ok ? j->msg : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, code, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_async_work(env, j->work);
    free(j);
}

static napi_value BatchAsync(napi_env env, napi_callback_info info) {
    size_t argc = 10; napi_value argv[10];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 8) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "batchAsync(handle, pcm, S, L, fftSize, hop, reassign, outDb[, outRgba, outIndex])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* pcm; size_t plen;
    if (!get_typed(env, argv[1], napi_float32_array, &pcm, &plen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm must be a Float32Array"); return NULL; }
    int32_t S, n, hop; int64_t L; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[3], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[6], &argv[6]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[6], &reassign));
    if (S < 1 || L < 1 || (size_t)S * (size_t)L != plen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm.length must equal S*L"); return NULL; }
    size_t l0 = 0, l1 = 0, l2 = 0; void *p0 = NULL, *p1 = NULL, *p2 = NULL;
    if (!get_typed(env, argv[7], napi_float32_array, &p0, &l0, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 8 && !get_typed(env, argv[8], napi_uint8_array, &p1, &l1, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 9 && !get_typed(env, argv[9], napi_uint8_array, &p2, &l2, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outIndex must be a Uint8Array"); return NULL; }
    int64_t C = emspec_num_columns(L, n, hop);
    size_t cells = p0 ? l0 : (p1 ? l1 / 4 : l2);
    if (C <= 0 || cells != (size_t)S * (size_t)C * (size_t)h->rows || (p1 && l1 != 4 * cells) || (p2 && l2 != cells) || (p0 && l0 != cells)) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "output arrays must hold exactly S*columns*rows cells, rows = the engine's row count (rgba: 4 bytes per cell)");
        return NULL;
    }
    batch_job* j = (batch_job*)calloc(1, sizeof(batch_job));
    if (!j) { napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "calloc"); return NULL; }
    j->e = h->e; j->pcm = (const float*)pcm; j->S = S; j->L = L; j->n = n; j->hop = hop; j->reassign = reassign ? 1 : 0; j->C = C;
    j->out.db = (float*)p0; j->out.rgba = (uint8_t*)p1; j->out.index = (uint8_t*)p2;
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j); napi_throw_error(env, "EMSPEC_NAPI", "napi_create_promise"); return NULL; }
    napi_create_reference(env, argv[1], 1, &j->refs[0]);                       /* keep the buffers alive */
    if (p0) napi_create_reference(env, argv[7], 1, &j->refs[1]);
    if (p1) napi_create_reference(env, argv[8], 1, &j->refs[2]);
    if (p2) napi_create_reference(env, argv[9], 1, &j->refs[3]);
    napi_create_string_utf8(env, "emspec.batchAsync", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, batch_execute, batch_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        for (int i = 0; i < 4; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
        free(j);
        napi_throw_error(env, "EMSPEC_NAPI", "could not queue async work");
        return NULL;
    }
    return promise;
}

/* batchPackedAsync(handle, pcm, S, L, fftSize, hop, reassign, wire, offsets:Float64Array(S+1)) -> Promise<columns>: batchPacked on the
 * libuv thread pool.  The typed arrays are kept alive by references; the caller touches neither them nor the engine before
 * the promise settles. */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[3];
    emspec_engine* e;
    const float* pcm;
    uint8_t* wire;
    int64_t wire_len;
    double* offs;
    int64_t* o64;
    int32_t S, n, hop, reassign;
    int64_t L, C;
    int rc;
    char msg[256];
} packed_job;

static void packed_execute(napi_env env, void* data) {
    (void)env;
    packed_job* j = (packed_job*)data;
    j->rc = emspec_batch_packed(j->e, j->pcm, j->S, j->L, j->n, j->hop, j->reassign, j->wire, j->wire_len, j->o64);
    if (j->rc != EMSPEC_OK) { strncpy(j->msg, emspec_last_error(j->e), sizeof(j->msg) - 1); j->msg[sizeof(j->msg) - 1] = 0; }
}

static void packed_complete(napi_env env, napi_status status, void* data) {
    packed_job* j = (packed_job*)data;
    if (status == napi_ok && j->rc == EMSPEC_OK) for (int32_t i = 0; i <= j->S; ++i) j->offs[i] = (double)j->o64[i];   /* (the array is still referenced) */
    for (int i = 0; i < 3; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
    if (status == napi_ok && j->rc == EMSPEC_OK) {
        napi_value v; napi_create_int64(env, j->C, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value code, msg, err;
        napi_create_string_utf8(env, status == napi_ok ? status_name(j->rc) : "EMSPEC_NAPI", NAPI_AUTO_LENGTH, &code);
        napi_create_string_utf8(env, status == napi_ok ? j->msg : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, code, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_async_work(env, j->work);
    free(j->o64);
    free(j);
}

static napi_value BatchPackedAsync(napi_env env, napi_callback_info info) {
    size_t argc = 9; napi_value argv[9];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 9) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "batchPackedAsync(handle, pcm, S, L, fftSize, hop, reassign, wire, offsets)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* pcm; size_t plen;
    if (!get_typed(env, argv[1], napi_float32_array, &pcm, &plen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm must be a Float32Array"); return NULL; }
    int32_t S, n, hop; int64_t L; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[3], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[6], &argv[6]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[6], &reassign));
    if (S < 1 || L < 1 || (size_t)S * (size_t)L != plen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm.length must equal S*L"); return NULL; }
    void *wire = NULL, *offs = NULL; size_t wlen = 0, olen = 0;
    if (!get_typed(env, argv[7], napi_uint8_array, &wire, &wlen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "wire must be a Uint8Array"); return NULL; }
    if (!get_typed(env, argv[8], napi_float64_array, &offs, &olen, 0) || olen != (size_t)S + 1) {
        napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "offsets must be a Float64Array(S + 1)"); return NULL;
    }
    packed_job* j = (packed_job*)calloc(1, sizeof(packed_job));
    if (j) j->o64 = (int64_t*)malloc(sizeof(int64_t) * ((size_t)S + 1));
    if (!j || !j->o64) { free(j); napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "out of host memory"); return NULL; }
    j->e = h->e; j->pcm = (const float*)pcm; j->S = S; j->L = L; j->n = n; j->hop = hop; j->reassign = reassign ? 1 : 0;
    j->C = emspec_num_columns(L, n, hop); j->wire = (uint8_t*)wire; j->wire_len = (int64_t)wlen; j->offs = (double*)offs;
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j->o64); free(j); napi_throw_error(env, "EMSPEC_NAPI", "napi_create_promise"); return NULL; }
    napi_create_reference(env, argv[1], 1, &j->refs[0]);
    napi_create_reference(env, argv[7], 1, &j->refs[1]);
    napi_create_reference(env, argv[8], 1, &j->refs[2]);
    napi_create_string_utf8(env, "emspec.batchPackedAsync", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, packed_execute, packed_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        for (int i = 0; i < 3; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
        free(j->o64); free(j);
        napi_throw_error(env, "EMSPEC_NAPI", "could not queue async work");
        return NULL;
    }
    return promise;
}

/* ---- live multi-stream streaming (emspec_columns / emspec_push_samples_multi): S streams per call, one launch ---- */

/* optional Float64Array(S) that receives int64 values (column indices / counts: exact in a double) */
static int get_f64_out(napi_env env, napi_value v, size_t want, double** out) {
    void* d = NULL; size_t len = 0;
    *out = NULL;
    if (!get_typed(env, v, napi_float64_array, &d, &len, 1)) return 0;
    if (d && len != want) return 0;
    *out = (double*)d;
    return 1;
}

/* columns(handle, frames:Float32Array(S*fftSize), S, fftSize, hop, reassign, outDb?:Float32Array(S*rows),
 *         outRgba?:Uint8Array(4*S*rows), outColumns?:Float64Array(S)) : one frame of each of S streams -> their finished columns */
static napi_value Columns(napi_env env, napi_callback_info info) {
    size_t argc = 9; napi_value argv[9];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 6) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "columns(handle, frames, S, fftSize, hop, reassign[, outDb, outRgba, outColumns])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* fr; size_t flen;
    if (!get_typed(env, argv[1], napi_float32_array, &fr, &flen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "frames must be a Float32Array"); return NULL; }
    int32_t S, n, hop; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[5], &argv[5]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[5], &reassign));
    if (S < 1 || n < 1 || (size_t)S * (size_t)n != flen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "frames.length must equal S*fftSize"); return NULL; }
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0; double* ocol = NULL;
    if (argc > 6 && !get_typed(env, argv[6], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 7 && !get_typed(env, argv[7], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 8 && !get_f64_out(env, argv[8], (size_t)S, &ocol)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outColumns must be a Float64Array(S)"); return NULL; }
    const size_t cells = (size_t)S * (size_t)h->rows;
    if ((db && dblen != cells) || (rgba && rgbalen != 4 * cells)) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must hold S*rows floats, outRgba 4*S*rows bytes (rows = the engine's row count)"); return NULL; }
    int64_t* c64 = ocol ? (int64_t*)malloc((size_t)S * sizeof(int64_t)) : NULL;
    if (ocol && !c64) { napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "malloc"); return NULL; }
    int rc = emspec_columns(h->e, (const float*)fr, S, n, hop, reassign ? 1 : 0, (float*)db, (uint8_t*)rgba, h->rows, c64);
    if (rc == EMSPEC_OK && ocol) for (int32_t s = 0; s < S; ++s) ocol[s] = (double)c64[s];
    free(c64);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

/* columnsFlush(handle, outDb?, outRgba?, outColumns?:Float64Array(S)): every stream with pending columns emits its next one */
static napi_value ColumnsFlush(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "columnsFlush(handle[, outDb, outRgba, outColumns])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    const int32_t S = emspec_live_streams(h->e);
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0; double* ocol = NULL;
    if (argc > 1 && !get_typed(env, argv[1], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 2 && !get_typed(env, argv[2], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 3 && !get_f64_out(env, argv[3], (size_t)S, &ocol)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outColumns must be a Float64Array(S)"); return NULL; }
    const size_t cells = (size_t)S * (size_t)h->rows;
    if ((db && dblen != cells) || (rgba && rgbalen != 4 * cells)) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must hold S*rows floats, outRgba 4*S*rows bytes (S = the live session's streams)"); return NULL; }
    int64_t* c64 = (ocol && S > 0) ? (int64_t*)malloc((size_t)S * sizeof(int64_t)) : NULL;
    int rc = emspec_columns_flush(h->e, (float*)db, (uint8_t*)rgba, h->rows, c64);
    if (rc == EMSPEC_OK && ocol && c64) for (int32_t s = 0; s < S; ++s) ocol[s] = (double)c64[s];
    free(c64);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

/* pushColumnsMulti(handle, count, fftSize, hop, reassign) -> the largest per-stream column count a block of `count` samples completes */
static napi_value PushColumnsMulti(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 5) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pushColumnsMulti(handle, count, fftSize, hop, reassign)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    int64_t count; int32_t n, hop; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[1], &count));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[4], &argv[4]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[4], &reassign));
    const int64_t k = emspec_push_columns_multi(h->e, count, n, hop, reassign ? 1 : 0);
    if (k < 0) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "invalid count / fftSize / hop"); return NULL; }
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, k, &r));
    return r;
}

/* pushMulti(handle, samples:Float32Array(S*count), S, fftSize, hop, reassign, maxColumns, outDb?:Float32Array(S*maxColumns*rows),
 *           outRgba?:Uint8Array(4*S*maxColumns*rows), outCounts?:Float64Array(S), outFirst?:Float64Array(S)) */
static napi_value PushMulti(napi_env env, napi_callback_info info) {
    size_t argc = 11; napi_value argv[11];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 7) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pushMulti(handle, samples, S, fftSize, hop, reassign, maxColumns[, outDb, outRgba, outCounts, outFirst])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* smp; size_t slen;
    if (!get_typed(env, argv[1], napi_float32_array, &smp, &slen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "samples must be a Float32Array"); return NULL; }
    int32_t S, n, hop; int64_t maxc; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[5], &argv[5]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[5], &reassign));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[6], &maxc));
    if (S < 1 || slen % (size_t)S != 0) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "samples.length must be a multiple of S"); return NULL; }
    if (maxc < 0) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "maxColumns must be >= 0"); return NULL; }
    const int64_t count = (int64_t)(slen / (size_t)S);
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0; double *ocnt = NULL, *ofirst = NULL;
    if (argc > 7 && !get_typed(env, argv[7], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 8 && !get_typed(env, argv[8], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 9 && !get_f64_out(env, argv[9], (size_t)S, &ocnt)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outCounts must be a Float64Array(S)"); return NULL; }
    if (argc > 10 && !get_f64_out(env, argv[10], (size_t)S, &ofirst)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outFirst must be a Float64Array(S)"); return NULL; }
    const size_t cells = (size_t)S * (size_t)maxc * (size_t)h->rows;
    if ((db && dblen != cells) || (rgba && rgbalen != 4 * cells)) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must hold S*maxColumns*rows floats, outRgba 4x that in bytes"); return NULL; }
    int64_t* tmp = (int64_t*)malloc((size_t)S * 2 * sizeof(int64_t));
    if (!tmp) { napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "malloc"); return NULL; }
    int rc = emspec_push_samples_multi(h->e, (const float*)smp, S, count, count, n, hop, reassign ? 1 : 0, (float*)db, (uint8_t*)rgba,
                                       h->rows, maxc, tmp, tmp + S);
    if (rc == EMSPEC_OK) for (int32_t s = 0; s < S; ++s) { if (ocnt) ocnt[s] = (double)tmp[s]; if (ofirst) ofirst[s] = (double)tmp[S + s]; }
    free(tmp);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

static napi_value ResetStream(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 2) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "resetStream(handle, stream)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    int32_t s;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[1], &s));
    int rc = emspec_reset_stream(h->e, s);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

static napi_value LiveStreams(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int32(env, emspec_live_streams(h->e), &r));
    return r;
}

/* columnsAsync(handle, frames, S, fftSize, hop, reassign, outDb?, outRgba?, outColumns?) -> Promise<undefined>: columns() on the
 * libuv thread pool (napi_create_async_work).  The typed arrays are kept alive by references; the caller touches neither them
 * nor the engine before the promise settles (an engine is not thread-safe). */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[4];
    emspec_engine* e;
    const float* frames;
    float* db; uint8_t* rgba; double* ocol; int64_t* c64;
    int32_t S, n, hop, reassign, rows;
    int rc;
    char msg[256];
} columns_job;

static void columns_execute(napi_env env, void* data) {
    (void)env;
    columns_job* j = (columns_job*)data;
    j->rc = emspec_columns(j->e, j->frames, j->S, j->n, j->hop, j->reassign, j->db, j->rgba, j->rows, j->c64);
    if (j->rc != EMSPEC_OK) { strncpy(j->msg, emspec_last_error(j->e), sizeof(j->msg) - 1); j->msg[sizeof(j->msg) - 1] = 0; }
}

static void columns_complete(napi_env env, napi_status status, void* data) {
    columns_job* j = (columns_job*)data;
    if (status == napi_ok && j->rc == EMSPEC_OK) {
        if (j->ocol) for (int32_t s = 0; s < j->S; ++s) j->ocol[s] = (double)j->c64[s];
        napi_value v; napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value code, msg, err;
        napi_create_string_utf8(env, status == napi_ok ? status_name(j->rc) : "EMSPEC_NAPI", NAPI_AUTO_LENGTH, &code);
        napi_create_string_utf8(env, status == napi_ok ? j->msg : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, code, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    for (int i = 0; i < 4; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
    napi_delete_async_work(env, j->work);
    free(j->c64);
    free(j);
}

static napi_value ColumnsAsync(napi_env env, napi_callback_info info) {
    size_t argc = 9; napi_value argv[9];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 6) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "columnsAsync(handle, frames, S, fftSize, hop, reassign[, outDb, outRgba, outColumns])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* fr; size_t flen;
    if (!get_typed(env, argv[1], napi_float32_array, &fr, &flen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "frames must be a Float32Array"); return NULL; }
    int32_t S, n, hop; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[5], &argv[5]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[5], &reassign));
    if (S < 1 || n < 1 || (size_t)S * (size_t)n != flen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "frames.length must equal S*fftSize"); return NULL; }
    void *db = NULL, *rgba = NULL; size_t dblen = 0, rgbalen = 0; double* ocol = NULL;
    if (argc > 6 && !get_typed(env, argv[6], napi_float32_array, &db, &dblen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    if (argc > 7 && !get_typed(env, argv[7], napi_uint8_array, &rgba, &rgbalen, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outRgba must be a Uint8Array"); return NULL; }
    if (argc > 8 && !get_f64_out(env, argv[8], (size_t)S, &ocol)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outColumns must be a Float64Array(S)"); return NULL; }
    const size_t cells = (size_t)S * (size_t)h->rows;
    if ((db && dblen != cells) || (rgba && rgbalen != 4 * cells)) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must hold S*rows floats, outRgba 4*S*rows bytes (rows = the engine's row count)"); return NULL; }
    columns_job* j = (columns_job*)calloc(1, sizeof(columns_job));
    if (!j) { napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "calloc"); return NULL; }
    j->c64 = (int64_t*)malloc((size_t)S * sizeof(int64_t));
    if (!j->c64) { free(j); napi_throw_error(env, "EMSPEC_ERR_OUT_OF_MEMORY", "malloc"); return NULL; }
    j->e = h->e; j->frames = (const float*)fr; j->db = (float*)db; j->rgba = (uint8_t*)rgba; j->ocol = ocol;
    j->S = S; j->n = n; j->hop = hop; j->reassign = reassign ? 1 : 0; j->rows = h->rows;
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j->c64); free(j); napi_throw_error(env, "EMSPEC_NAPI", "napi_create_promise"); return NULL; }
    napi_create_reference(env, argv[1], 1, &j->refs[0]);
    if (db) napi_create_reference(env, argv[6], 1, &j->refs[1]);
    if (rgba) napi_create_reference(env, argv[7], 1, &j->refs[2]);
    if (ocol) napi_create_reference(env, argv[8], 1, &j->refs[3]);
    napi_create_string_utf8(env, "emspec.columnsAsync", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, columns_execute, columns_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        for (int i = 0; i < 4; ++i) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
        free(j->c64); free(j);
        napi_throw_error(env, "EMSPEC_NAPI", "could not queue async work");
        return NULL;
    }
    return promise;
}

static napi_value SetColormap(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* lut; size_t len;
    if (argc < 2 || !get_typed(env, argv[1], napi_uint8_array, &lut, &len, 0) || len != 1024) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "colormap must be a Uint8Array(1024) of RGBA"); return NULL; }
    int rc = emspec_set_colormap(h->e, (const uint8_t*)lut);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

static napi_value SetDisplay(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    double sm = 0, agc = 0;
    if (argc < 3 || napi_get_value_double(env, argv[1], &sm) != napi_ok || napi_get_value_double(env, argv[2], &agc) != napi_ok) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "setDisplay(handle, smoothing, agcStrength)"); return NULL; }
    int rc = emspec_set_display(h->e, (float)sm, (float)agc);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

/* setRowEdges(handle, Float32Array(rows+1) | null) ; getRowEdges(handle, Float32Array(rows+1)) */
static napi_value SetRowEdges(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* p = NULL; size_t len = 0;
    if (argc < 2 || !get_typed(env, argv[1], napi_float32_array, &p, &len, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "edges must be a Float32Array(rows+1) or null"); return NULL; }
    int rc = emspec_set_row_edges_hz(h->e, (const float*)p, (int32_t)len);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}
static napi_value GetRowEdges(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* p = NULL; size_t len = 0;
    if (argc < 2 || !get_typed(env, argv[1], napi_float32_array, &p, &len, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "out must be a Float32Array(rows+1)"); return NULL; }
    int rc = emspec_get_row_edges_hz(h->e, (float*)p, (int32_t)len);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}

/* allocPinned(byteLength) -> ArrayBuffer backed by page-locked host memory (emspec_host_alloc);
 * typed arrays over it make batch()/column() copies run at full PCIe speed.  Freed by the GC. */
static void finalize_pinned(napi_env env, void* data, void* hint) { (void)env; (void)hint; emspec_host_free(data); }
static napi_value AllocPinned(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int64_t bytes = 0;
    if (argc < 1 || napi_get_value_int64(env, argv[0], &bytes) != napi_ok || bytes <= 0) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "allocPinned(byteLength > 0)"); return NULL; }
    void* p = NULL;
    int rc = emspec_host_alloc((size_t)bytes, &p);
    if (rc != EMSPEC_OK) { napi_throw_error(env, status_name(rc), "page-locked allocation failed"); return NULL; }
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, finalize_pinned, NULL, &ab) != napi_ok) {
        emspec_host_free(p);
        napi_throw_error(env, "EMSPEC_NAPI", "napi_create_external_arraybuffer failed");
        return NULL;
    }
    return ab;
}

static napi_value NumColumns(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int64_t L = 0; int32_t n = 0, hop = 0;
    if (argc < 3) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "numColumns(L, fftSize, hop)"); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[0], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[1], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &hop));
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, emspec_num_columns(L, n, hop), &r));
    return r;
}

static napi_value LatencyColumns(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    int32_t n = 0, hop = 0; bool re = true;
    if (argc < 2) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "latencyColumns(fftSize, hop[, reassign])"); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[0], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[1], &hop));
    if (argc > 2) { NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[2], &argv[2])); NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[2], &re)); }
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int32(env, emspec_latency_columns(n, hop, re ? 1 : 0), &r));
    return r;
}

/* ---- multi-GPU (include/emspec.h "Multi-GPU"): one node process per GPU ---- */
/* commUniqueId() -> Uint8Array(128): rank 0 calls it and hands the bytes to the other rank processes (IPC, a file...) */
static napi_value CommUniqueId(napi_env env, napi_callback_info info) {
    (void)info;
    void* data; napi_value ab, ta;
    NAPI_OK_OR_RETURN(env, napi_create_arraybuffer(env, EMSPEC_COMM_ID_BYTES, &data, &ab));
    if (emspec_comm_unique_id((uint8_t*)data) != EMSPEC_OK) { napi_throw_error(env, "EMSPEC_ERR_COMM", emspec_last_error(NULL)); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_create_typedarray(env, napi_uint8_array, EMSPEC_COMM_ID_BYTES, ab, 0, &ta));
    return ta;
}
/* commInit(handle, id:Uint8Array(128), rank, world): collective over all rank processes */
static napi_value CommInit(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 4) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "commInit(handle, id, rank, world)"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* id; size_t len; int32_t rank, world;
    if (!get_typed(env, argv[1], napi_uint8_array, &id, &len, 0) || len != EMSPEC_COMM_ID_BYTES) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "id must be the Uint8Array(128) of commUniqueId()"); return NULL; }
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &rank));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[3], &world));
    int rc = emspec_comm_init(h->e, (const uint8_t*)id, rank, world);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    return NULL;
}
/* batchGather(handle, pcm, S, L, fftSize, hop, reassign, root, outAllIndex?:Uint8Array(world*S*C*rows), outDb?:Float32Array(S*C*rows))
 * -> bytes this rank put on the wire.  Collective: every rank process calls it with its own shard of the streams. */
static napi_value BatchGather(napi_env env, napi_callback_info info) {
    size_t argc = 10; napi_value argv[10];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 8) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "batchGather(handle, pcm, S, L, fftSize, hop, reassign, root[, outAllIndex, outDb])"); return NULL; }
    handle_t* h = get_handle(env, argv[0]); if (!h) return NULL;
    void* pcm; size_t plen;
    if (!get_typed(env, argv[1], napi_float32_array, &pcm, &plen, 0)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm must be a Float32Array"); return NULL; }
    int32_t S, n, hop, root; int64_t L; bool reassign;
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[2], &S));
    NAPI_OK_OR_RETURN(env, napi_get_value_int64(env, argv[3], &L));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[4], &n));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[5], &hop));
    NAPI_OK_OR_RETURN(env, napi_coerce_to_bool(env, argv[6], &argv[6]));
    NAPI_OK_OR_RETURN(env, napi_get_value_bool(env, argv[6], &reassign));
    NAPI_OK_OR_RETURN(env, napi_get_value_int32(env, argv[7], &root));
    if (S < 1 || L < 1 || (size_t)S * (size_t)L != plen) { napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "pcm.length must equal S*L"); return NULL; }
    void *pall = NULL, *pdb = NULL; size_t lall = 0, ldb = 0;
    if (argc > 8 && !get_typed(env, argv[8], napi_uint8_array, &pall, &lall, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outAllIndex must be a Uint8Array"); return NULL; }
    if (argc > 9 && !get_typed(env, argv[9], napi_float32_array, &pdb, &ldb, 1)) { napi_throw_type_error(env, "EMSPEC_ERR_INVALID_ARG", "outDb must be a Float32Array"); return NULL; }
    const int64_t C = emspec_num_columns(L, n, hop);
    const int32_t world = emspec_comm_world(h->e), rank = emspec_comm_rank(h->e);
    const size_t cells = (size_t)S * (size_t)(C > 0 ? C : 0) * (size_t)h->rows;
    if (world < 1) { napi_throw_error(env, "EMSPEC_ERR_STATE", "no communicator: call commInit first"); return NULL; }
    if (C <= 0 || (rank == root && lall != cells * (size_t)world) || (pdb && ldb != cells)) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "outAllIndex must hold world*S*columns*rows bytes on the root, outDb S*columns*rows floats");
        return NULL;
    }
    int64_t sent = 0;
    int rc = emspec_batch_gather(h->e, (const float*)pcm, S, L, n, hop, reassign ? 1 : 0, root, rank == root ? (uint8_t*)pall : NULL,
                                 (float*)pdb, &sent);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_create_int64(env, sent, &r));
    return r;
}

/* buildInfo() -> "emspec abi=2 sources=<sha16> arch=gfx950" */
static napi_value BuildInfo(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value r;
    NAPI_OK_OR_RETURN(env, napi_create_string_utf8(env, emspec_build_info(), NAPI_AUTO_LENGTH, &r));
    return r;
}

/* deviceStatus(handle): synchronise the device; throws when a kernel flagged a protocol error */
static napi_value DeviceStatus(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK_OR_RETURN(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t* h = NULL;
    if (argc < 1 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !h || !h->e) {
        napi_throw_error(env, "EMSPEC_ERR_INVALID_ARG", "engine handle expected");
        return NULL;
    }
    int rc = emspec_device_status(h->e);
    if (rc != EMSPEC_OK) return throw_status(env, h->e, rc);
    napi_value r; NAPI_OK_OR_RETURN(env, napi_get_undefined(env, &r));
    return r;
}

static napi_value Init(napi_env env, napi_value exports) {
    napi_property_descriptor props[] = {
        {"create", NULL, Create, NULL, NULL, NULL, napi_default, NULL},
        {"destroy", NULL, Destroy, NULL, NULL, NULL, napi_default, NULL},
        {"rows", NULL, Rows, NULL, NULL, NULL, napi_default, NULL},
        {"column", NULL, Column, NULL, NULL, NULL, napi_default, NULL},
        {"flush", NULL, Flush, NULL, NULL, NULL, napi_default, NULL},
        {"push", NULL, Push, NULL, NULL, NULL, napi_default, NULL},
        {"pushColumns", NULL, PushColumns, NULL, NULL, NULL, napi_default, NULL},
        {"warpedEdges", NULL, WarpedEdges, NULL, NULL, NULL, napi_default, NULL},
        {"referenceColormap", NULL, ReferenceColormap, NULL, NULL, NULL, napi_default, NULL},
        {"reset", NULL, Reset, NULL, NULL, NULL, napi_default, NULL},
        {"batch", NULL, Batch, NULL, NULL, NULL, napi_default, NULL},
        {"batchAsync", NULL, BatchAsync, NULL, NULL, NULL, napi_default, NULL},
        {"batchPacked", NULL, BatchPacked, NULL, NULL, NULL, napi_default, NULL},
        {"batchPackedAsync", NULL, BatchPackedAsync, NULL, NULL, NULL, napi_default, NULL},
        {"wireUnpack", NULL, WireUnpack, NULL, NULL, NULL, napi_default, NULL},
        {"wireBound", NULL, WireBound, NULL, NULL, NULL, napi_default, NULL},
        {"setColormap", NULL, SetColormap, NULL, NULL, NULL, napi_default, NULL},
        {"setDisplay", NULL, SetDisplay, NULL, NULL, NULL, napi_default, NULL},
        {"setRowEdges", NULL, SetRowEdges, NULL, NULL, NULL, napi_default, NULL},
        {"getRowEdges", NULL, GetRowEdges, NULL, NULL, NULL, napi_default, NULL},
        {"allocPinned", NULL, AllocPinned, NULL, NULL, NULL, napi_default, NULL},
        {"numColumns", NULL, NumColumns, NULL, NULL, NULL, napi_default, NULL},
        {"latencyColumns", NULL, LatencyColumns, NULL, NULL, NULL, napi_default, NULL},
        {"commUniqueId", NULL, CommUniqueId, NULL, NULL, NULL, napi_default, NULL},
        {"commInit", NULL, CommInit, NULL, NULL, NULL, napi_default, NULL},
        {"batchGather", NULL, BatchGather, NULL, NULL, NULL, napi_default, NULL},
        {"buildInfo", NULL, BuildInfo, NULL, NULL, NULL, napi_default, NULL},
        {"deviceStatus", NULL, DeviceStatus, NULL, NULL, NULL, napi_default, NULL},
        {"columns", NULL, Columns, NULL, NULL, NULL, napi_default, NULL},
        {"columnsAsync", NULL, ColumnsAsync, NULL, NULL, NULL, napi_default, NULL},
        {"columnsFlush", NULL, ColumnsFlush, NULL, NULL, NULL, napi_default, NULL},
        {"pushMulti", NULL, PushMulti, NULL, NULL, NULL, napi_default, NULL},
        {"pushColumnsMulti", NULL, PushColumnsMulti, NULL, NULL, NULL, napi_default, NULL},
        {"resetStream", NULL, ResetStream, NULL, NULL, NULL, napi_default, NULL},
        {"liveStreams", NULL, LiveStreams, NULL, NULL, NULL, napi_default, NULL},
    };
    napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
