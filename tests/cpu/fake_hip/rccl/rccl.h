// TEST INFRASTRUCTURE: the few RCCL types csrc/rtd_api.hip names, for the host-only sanitizer build (see ../hip/hip_runtime.h).
#pragma once
#include <cstddef>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
