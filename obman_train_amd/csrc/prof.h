#pragma once
#include <hip/hip_runtime.h>

// kernel ids for obman_prof_summary()
enum {
  OBMAN_K_PAIRMIN_FWD = 1, OBMAN_K_PAIRMIN_BWD = 2, OBMAN_K_CONTAINS = 3, OBMAN_K_CONTACT_FWD = 4,
  OBMAN_K_CONTACT_BWD = 5, OBMAN_K_MANO_FWD = 6, OBMAN_K_MANO_BWD = 7, OBMAN_K_DECODER_FWD = 8,
  OBMAN_K_DECODER_BWD = 9,
  OBMAN_K_CHAMFER_FWD = 10, OBMAN_K_CHAMFER_BWD = 11,  // ChamferLoss launches only (1/2 = hand<->object closest-vertex launches)
  // asymmetric sizes run one launch per direction: the y -> x direction's launch is booked separately (x -> y, merged and
  // single-launch forms stay under OBMAN_K_CHAMFER_FWD / OBMAN_K_PAIRMIN_FWD)
  OBMAN_K_CHAMFER_FWD_Y = 12, OBMAN_K_PAIRMIN_FWD_Y = 13,
};

struct ObmanProfScope {
  ObmanProfScope(int id, hipStream_t st);
  ~ObmanProfScope();
  int rec_;
  hipStream_t st_;
};
