// Translation unit of k_augru_x alone (the AUGRU recurrence of the DIEN scorer in fp16x2 form, augru_x.hpp).
//
// Why its own unit: dien.hip is built with -fno-slp-vectorize (the VALU epilogues of k_cat_attn2 / k_din_x / k_gru_h16 run beside
// another wave's MFMAs, and packed fp32 VALU serialises with the matrix pipe: tools/mfma_valu_overlap.hip), but under that flag
// the reward-sized form k_augru_x<2,4,2> needs 256 registers + 8 spilled (36 B of scratch) where the default pipeline fits it in
// 254 without scratch (tools/codeobj_notes.py).  This unit keeps the default; dien.hip launches through the two
// functions below.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "recur_args.hpp"
#include "augru_x.hpp"

namespace rl4rs {

// raise the dynamic-LDS limit of both row-tile forms (once per device: raise_dyn_smem); RL4RS_OK or an error code
int augru_x_prepare() {
    int rc;
    if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_augru_x<1, RL4RS_X_NRES, RL4RS_X_RING>), augru_x_smem(1)))) return rc;
    if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_augru_x<2, RL4RS_X2_NRES, RL4RS_X2_RING>), augru_x_smem(2)))) return rc;
    if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_augru_x<1, RL4RS_X_NRES, RL4RS_X_RING, true>), augru_x_smem(1)))) return rc;
    return 0;
}

// rows_per_wg = 32: grid (ceil(n_rows / 32), S);  64: grid (n_rows / 64, S) - the caller has checked the 64-row form's conditions
void augru_x_launch(int rows_per_wg, int n_seq, hipStream_t st, const RecurArgs& a) {
    const dim3 block(512);
    if (rows_per_wg == 64)      // (the 64-row form with the redirect - 256 registers, no spill - measured flat on SeqSlate and 0.3 % slower on the headline: not instantiated)
        hipLaunchKernelGGL((k_augru_x<2, RL4RS_X2_NRES, RL4RS_X2_RING>), dim3(a.n_rows / 64, n_seq), block, augru_x_smem(2), st, a);
    else if (a.lead[0])       // the handle keeps a pad slot and the leading-zero counts of its cache slots (RecurArgs::lead)
        hipLaunchKernelGGL((k_augru_x<1, RL4RS_X_NRES, RL4RS_X_RING, true>), dim3((a.n_rows + 31) / 32, n_seq), block, augru_x_smem(1), st, a);
    else
        hipLaunchKernelGGL((k_augru_x<1, RL4RS_X_NRES, RL4RS_X_RING>), dim3((a.n_rows + 31) / 32, n_seq), block, augru_x_smem(1), st, a);
}

}  // namespace rl4rs
