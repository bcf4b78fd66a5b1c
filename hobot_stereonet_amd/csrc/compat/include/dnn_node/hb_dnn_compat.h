// hb_dnn_compat.h — the subset of Horizon's libdnn C types / functions that hobot_stereonet touches,
// re-implemented over host memory + libstereonet_hip.so.  Inferred from the reference's call sites only
// (the real hb_dnn.h is not vendored): stereonet_infer/src/preprocess.cpp:54-65,81-92,952-973,1052-1055,
// stereonet_infer/src/parser.cpp:35,60,169-188, stereonet_infer/src/stereonet_node.cpp:57-103.
#pragma once
#include <cstdint>

extern "C" {

typedef void* hbDNNHandle_t;

typedef struct {
  uint64_t phyAddr;   // unused on this platform (no BPU-visible physical memory); kept for layout parity
  void* virAddr;
  uint32_t memSize;
} hbSysMem;

typedef enum { HB_DNN_LAYOUT_NHWC = 0, HB_DNN_LAYOUT_NCHW = 2, HB_DNN_LAYOUT_NONE = 255 } hbDNNTensorLayout;

typedef enum {
  HB_DNN_TENSOR_TYPE_S8 = 10,
  HB_DNN_TENSOR_TYPE_U8 = 11,
  HB_DNN_TENSOR_TYPE_S32 = 14,
  HB_DNN_TENSOR_TYPE_F32 = 13
} hbDNNDataType;

typedef enum { HB_SYS_MEM_CACHE_INVALIDATE = 1, HB_SYS_MEM_CACHE_CLEAN = 2 } hbSysMemFlushFlag;

#define HB_DNN_TENSOR_MAX_DIMENSIONS 8
typedef struct {
  int32_t dimensionSize[HB_DNN_TENSOR_MAX_DIMENSIONS];
  int32_t numDimensions;
} hbDNNTensorShape;

typedef struct {
  int32_t scaleLen;
  float* scaleData;
  int32_t zeroPointLen;
  int8_t* zeroPointData;
} hbDNNQuantiScale;

typedef struct {
  hbDNNTensorShape validShape;
  hbDNNTensorShape alignedShape;
  int32_t tensorLayout;
  int32_t tensorType;
  hbDNNQuantiScale scale;
  int32_t alignedByteSize;
} hbDNNTensorProperties;

// host allocations; "flush" is a no-op (the engine copies through pinned staging, sn_submit)
int32_t hbSysAllocCachedMem(hbSysMem* mem, uint32_t size);
int32_t hbSysAllocMem(hbSysMem* mem, uint32_t size);
int32_t hbSysFreeMem(hbSysMem* mem);
int32_t hbSysFlushMem(hbSysMem* mem, int32_t flag);

int32_t hbDNNGetInputTensorProperties(hbDNNTensorProperties* properties, hbDNNHandle_t dnnHandle, int32_t inputIndex);
int32_t hbDNNGetOutputTensorProperties(hbDNNTensorProperties* properties, hbDNNHandle_t dnnHandle, int32_t outputIndex);

}  // extern "C"
