// Internal: fast path of stg_attn_fwd / stg_attn_bwd for long single-head K == V attention (xattn.hip).
#pragma once
#include "../../include/stgcma.h"
bool stg_xattn_eligible(const stg_attn_args* f, bool need_lse);
int stg_xattn_fwd(const stg_attn_args* f, void* stream);
int stg_xattn_bwd(const stg_attn_bwd_args* b, void* stream);   // requires b->dV == NULL (dK receives dK + dV)
bool stg_xattn_pairable(const stg_attn_args* f0, const stg_attn_args* f1);
int stg_xattn_fwd2(const stg_attn_args* f0, const stg_attn_args* f1, void* stream);       // both eligible and pairable: one launch, grid.y = 2
int stg_xattn_bwd2(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* stream);
