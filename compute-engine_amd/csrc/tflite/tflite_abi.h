// Minimal mirror of the TensorFlow Lite C API pieces that Larq Compute Engine's custom
// ops touch (tensorflow v2.16.1: tensorflow/lite/core/c/common.h, c_api_types.h,
// builtin_op_data.h).  TensorFlow is an un-vendored dependency of the reference
// (third_party/tensorflow is an empty submodule) and is not in this image, so the op glue
// is compiled against this header: same type names, field order and enum values, written
// from the published API.  When building inside a real TFLite tree, compile with
// -DLCE_USE_SYSTEM_TFLITE to include TFLite's own headers instead.
#pragma once

#ifdef LCE_USE_SYSTEM_TFLITE
#include "tensorflow/lite/core/c/builtin_op_data.h"
#include "tensorflow/lite/core/c/common.h"
#else

#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum TfLiteStatus {
  kTfLiteOk = 0, kTfLiteError = 1, kTfLiteDelegateError = 2, kTfLiteApplicationError = 3,
  kTfLiteDelegateDataNotFound = 4, kTfLiteDelegateDataWriteError = 5,
  kTfLiteDelegateDataReadError = 6, kTfLiteUnresolvedOps = 7, kTfLiteCancelled = 8,
} TfLiteStatus;

typedef enum {
  kTfLiteNoType = 0, kTfLiteFloat32 = 1, kTfLiteInt32 = 2, kTfLiteUInt8 = 3, kTfLiteInt64 = 4,
  kTfLiteString = 5, kTfLiteBool = 6, kTfLiteInt16 = 7, kTfLiteComplex64 = 8, kTfLiteInt8 = 9,
  kTfLiteFloat16 = 10, kTfLiteFloat64 = 11, kTfLiteComplex128 = 12, kTfLiteUInt64 = 13,
  kTfLiteResource = 14, kTfLiteVariant = 15, kTfLiteUInt32 = 16, kTfLiteUInt16 = 17,
  kTfLiteInt4 = 18,
} TfLiteType;

typedef struct TfLiteQuantizationParams {
  float scale;
  int32_t zero_point;
} TfLiteQuantizationParams;

typedef enum TfLiteQuantizationType { kTfLiteNoQuantization = 0, kTfLiteAffineQuantization = 1 } TfLiteQuantizationType;
typedef struct TfLiteQuantization {
  TfLiteQuantizationType type;
  void* params;
} TfLiteQuantization;

typedef struct TfLiteIntArray {
  int size;
  int data[];
} TfLiteIntArray;
typedef struct TfLiteFloatArray {
  int size;
  float data[];
} TfLiteFloatArray;
typedef struct TfLiteAffineQuantization {
  TfLiteFloatArray* scale;
  TfLiteIntArray* zero_point;
  int32_t quantized_dimension;
} TfLiteAffineQuantization;

// Provided by the TFLite runtime (tensorflow/lite/core/c/common.cc); the standalone test
// driver in this directory defines them too.
TfLiteIntArray* TfLiteIntArrayCreate(int size);
TfLiteIntArray* TfLiteIntArrayCopy(const TfLiteIntArray* src);
void TfLiteIntArrayFree(TfLiteIntArray* a);

typedef union TfLitePtrUnion {
  int32_t* i32; uint32_t* u32; int64_t* i64; uint64_t* u64; float* f; void* f16; double* f64;
  char* raw; const char* raw_const; uint8_t* uint8; bool* b; int16_t* i16; uint16_t* ui16;
  void* c64; void* c128; int8_t* int8; void* data;
} TfLitePtrUnion;

typedef enum TfLiteAllocationType {
  kTfLiteMemNone = 0, kTfLiteMmapRo, kTfLiteArenaRw, kTfLiteArenaRwPersistent, kTfLiteDynamic,
  kTfLitePersistentRo, kTfLiteCustom, kTfLiteVariantObject,
} TfLiteAllocationType;

typedef int TfLiteBufferHandle;
struct TfLiteDelegate;
struct TfLiteContext;
struct TfLiteSparsity;

typedef struct TfLiteTensor {
  TfLiteType type;
  TfLitePtrUnion data;
  TfLiteIntArray* dims;
  TfLiteQuantizationParams params;
  TfLiteAllocationType allocation_type;
  size_t bytes;
  const void* allocation;
  const char* name;
  struct TfLiteDelegate* delegate;
  TfLiteBufferHandle buffer_handle;
  bool data_is_stale;
  bool is_variable;
  TfLiteQuantization quantization;
  struct TfLiteSparsity* sparsity;
  const TfLiteIntArray* dims_signature;
} TfLiteTensor;

typedef struct TfLiteNode {
  TfLiteIntArray* inputs;
  TfLiteIntArray* outputs;
  TfLiteIntArray* intermediates;
  TfLiteIntArray* temporaries;
  void* user_data;
  void* builtin_data;
  const void* custom_initial_data;
  int custom_initial_data_size;
  struct TfLiteDelegate* delegate;
  bool might_have_side_effect;
} TfLiteNode;

#define kTfLiteOptionalTensor (-1)

typedef struct TfLiteContext {
  size_t tensors_size;
  TfLiteStatus (*GetExecutionPlan)(struct TfLiteContext*, TfLiteIntArray** execution_plan);
  TfLiteTensor* tensors;
  void* impl_;
  TfLiteStatus (*ResizeTensor)(struct TfLiteContext*, TfLiteTensor* tensor, TfLiteIntArray* new_size);
  void (*ReportError)(struct TfLiteContext*, const char* msg, ...);
  TfLiteStatus (*AddTensors)(struct TfLiteContext*, int tensors_to_add, int* first_new_tensor_index);
  TfLiteStatus (*GetNodeAndRegistration)(struct TfLiteContext*, int, TfLiteNode**, void**);
  TfLiteStatus (*ReplaceNodeSubsetsWithDelegateKernels)(struct TfLiteContext*, void*, const TfLiteIntArray*, struct TfLiteDelegate*);
  int recommended_num_threads;
  void* (*GetExternalContext)(struct TfLiteContext*, int);
  void (*SetExternalContext)(struct TfLiteContext*, int, void*);
  bool allow_fp32_relax_to_fp16;
  void* profiler;
  // (later fields of the real struct are never touched by the LCE ops)
} TfLiteContext;

typedef struct TfLiteRegistration {
  void* (*init)(TfLiteContext* context, const char* buffer, size_t length);
  void (*free)(TfLiteContext* context, void* buffer);
  TfLiteStatus (*prepare)(TfLiteContext* context, TfLiteNode* node);
  TfLiteStatus (*invoke)(TfLiteContext* context, TfLiteNode* node);
  const char* (*profiling_string)(const TfLiteContext* context, const TfLiteNode* node);
  int32_t builtin_code;
  const char* custom_name;
  int version;
  void* registration_external;
  void* async_kernel;
  uint64_t inplace_operator;
} TfLiteRegistration;

// tensorflow/lite/core/c/builtin_op_data.h
typedef enum { kTfLitePaddingUnknown = 0, kTfLitePaddingSame, kTfLitePaddingValid } TfLitePadding;
typedef struct {
  int width; int height; int width_offset; int height_offset;
} TfLitePaddingValues;
typedef enum {
  kTfLiteActNone = 0, kTfLiteActRelu, kTfLiteActReluN1To1, kTfLiteActRelu6, kTfLiteActTanh,
  kTfLiteActSignBit, kTfLiteActSigmoid,
} TfLiteFusedActivation;

#ifdef __cplusplus
}
#endif
#endif  // LCE_USE_SYSTEM_TFLITE

// ---------------------------------------------------------------------------------------------
// Layout table (LP64) of the TFLite v2.16.1 structures the LCE ops read or write
// (tensorflow/lite/core/c/common.h at the commit pinned by the reference's WORKSPACE:12-24).  The op glue
// is handed pointers to these by the interpreter, so field ORDER is the ABI: the asserts hold for the
// mirror above and -- with -DLCE_USE_SYSTEM_TFLITE -- are checked against TensorFlow's own header, so a
// TensorFlow bump that moves a field fails the build instead of corrupting tensors at run time.
// ---------------------------------------------------------------------------------------------
#ifdef __cplusplus
#define LCE_ABI_AT(T, field, off) static_assert(offsetof(T, field) == (off), "TFLite ABI drift: " #T "::" #field)
static_assert(sizeof(void*) == 8 && sizeof(size_t) == 8, "LP64 only");
LCE_ABI_AT(TfLiteTensor, type, 0);
LCE_ABI_AT(TfLiteTensor, data, 8);
LCE_ABI_AT(TfLiteTensor, dims, 16);
LCE_ABI_AT(TfLiteTensor, params, 24);
LCE_ABI_AT(TfLiteTensor, allocation_type, 32);
LCE_ABI_AT(TfLiteTensor, bytes, 40);
LCE_ABI_AT(TfLiteTensor, allocation, 48);
LCE_ABI_AT(TfLiteTensor, name, 56);
LCE_ABI_AT(TfLiteTensor, delegate, 64);
LCE_ABI_AT(TfLiteTensor, buffer_handle, 72);
LCE_ABI_AT(TfLiteTensor, data_is_stale, 76);
LCE_ABI_AT(TfLiteTensor, is_variable, 77);
LCE_ABI_AT(TfLiteTensor, quantization, 80);
LCE_ABI_AT(TfLiteTensor, sparsity, 96);
LCE_ABI_AT(TfLiteTensor, dims_signature, 104);
static_assert(sizeof(TfLiteTensor) == 112, "TFLite ABI drift: sizeof(TfLiteTensor) (context->tensors is indexed by it)");
static_assert(sizeof(TfLiteQuantizationParams) == 8 && sizeof(TfLiteQuantization) == 16 && sizeof(TfLitePtrUnion) == 8,
              "TFLite ABI drift: tensor member sizes");
LCE_ABI_AT(TfLiteAffineQuantization, scale, 0);
LCE_ABI_AT(TfLiteAffineQuantization, zero_point, 8);
LCE_ABI_AT(TfLiteAffineQuantization, quantized_dimension, 16);
LCE_ABI_AT(TfLiteIntArray, data, 4);
LCE_ABI_AT(TfLiteNode, inputs, 0);
LCE_ABI_AT(TfLiteNode, outputs, 8);
LCE_ABI_AT(TfLiteNode, intermediates, 16);
LCE_ABI_AT(TfLiteNode, temporaries, 24);
LCE_ABI_AT(TfLiteNode, user_data, 32);
LCE_ABI_AT(TfLiteNode, builtin_data, 40);
LCE_ABI_AT(TfLiteNode, custom_initial_data, 48);
LCE_ABI_AT(TfLiteNode, custom_initial_data_size, 56);
LCE_ABI_AT(TfLiteContext, tensors_size, 0);
LCE_ABI_AT(TfLiteContext, tensors, 16);
LCE_ABI_AT(TfLiteContext, ResizeTensor, 32);
LCE_ABI_AT(TfLiteContext, ReportError, 40);
LCE_ABI_AT(TfLiteContext, AddTensors, 48);
LCE_ABI_AT(TfLiteContext, recommended_num_threads, 72);
LCE_ABI_AT(TfLiteRegistration, init, 0);
LCE_ABI_AT(TfLiteRegistration, free, 8);
LCE_ABI_AT(TfLiteRegistration, prepare, 16);
LCE_ABI_AT(TfLiteRegistration, invoke, 24);
LCE_ABI_AT(TfLiteRegistration, profiling_string, 32);
LCE_ABI_AT(TfLiteRegistration, builtin_code, 40);
LCE_ABI_AT(TfLiteRegistration, custom_name, 48);
LCE_ABI_AT(TfLiteRegistration, version, 56);
static_assert(kTfLiteFloat32 == 1 && kTfLiteInt32 == 2 && kTfLiteBool == 6 && kTfLiteInt8 == 9, "TFLite ABI drift: TfLiteType");
static_assert(kTfLiteMmapRo == 1 && kTfLiteArenaRw == 2 && kTfLiteDynamic == 4, "TFLite ABI drift: TfLiteAllocationType");
static_assert(kTfLitePaddingSame == 1 && kTfLitePaddingValid == 2 && kTfLiteActRelu == 1 && kTfLiteActReluN1To1 == 2 &&
                  kTfLiteActRelu6 == 3, "TFLite ABI drift: builtin_op_data enums");
#undef LCE_ABI_AT
#endif
