// gram_mfma.hpp -- MFMA gradient-Gram element kernel (filled in below)
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "igx.hpp"
namespace igx {
static int try_gram_mfma(const Space &, const SpaceDev &, const OutDev &, hipStream_t, bool forced, std::string &, int &, std::string &err, bool &done) {
  done = false;
  if (forced) { err = "MFMA kernel does not cover this configuration"; return IGX_ERR_SUP; }
  return 0;
}
}
