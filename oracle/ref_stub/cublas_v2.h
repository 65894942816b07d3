// TEST INFRASTRUCTURE (oracle/ref_stub): the reference's tracker utils.h only names the cuBLAS status type.
#pragma once
typedef int cublasStatus_t;
enum { CUBLAS_STATUS_SUCCESS = 0 };
