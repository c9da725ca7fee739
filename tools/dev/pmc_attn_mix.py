"""The attention launches of one 70-frame clip (bench.py's mix: per propagated frame and layer one self-attention over the
frame and one long-term attention over the bank, M = 1 + (t-1)//5 memorised frames), on the default stream, for
rocprofv3 --pmc passes (PMC collection hangs on bench.py's per-clip HIP streams).
    python tools/dev/pmc_attn_mix.py [aot|aotx6|gated|gatedx6|m14]      aot: R50-AOTL (attn_fwd_d32_pipe_kernel; aotx6: attn_x6_d32_kernel), gated: R50-DeAOTL
    (attn_fwd_wide_coop_kernel<8>; gatedx6: attn_x6_wide64p_kernel), m14: three launches of each kernel -- fp32 and bf16x6 forms -- at M = 14 (SQ counter passes)
AOT_HIP_LIB selects a variant build of the library."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
from networks.layers.attention import attn_splits, gated_splits, gated_splits_x6, _planned_len
mode = sys.argv[1] if len(sys.argv) > 1 else 'aot'
N, C, H, E, CAP = 1674, 256, 8, 1024, 32
q = torch.randn(N, C, device='cuda'); out = torch.empty(N, C, device='cuda')
k = torch.randn(14 * N, C, device='cuda'); v = torch.randn(14 * N, C, device='cuda')
part = torch.empty(4 * N * (C + 2 * H), device='cuda')      # sized like MultiheadAttention.core's slab set
gq = torch.randn(N, 128, device='cuda'); gk = torch.randn(14 * N, 128, device='cuda'); gv = torch.randn(14 * N, E, device='cuda')
gu = torch.randn(N, E, device='cuda'); go = torch.empty(N, E, device='cuda'); gpart = torch.empty(16 * N * (E + 8), device='cuda')


def d32(T, brows):
    ns = attn_splits(N, H, _planned_len(T, N, brows), wg_waves=4)
    aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)


def gated(T, brows):
    ns = gated_splits(N, _planned_len(T, N, brows))
    aot_hip.gated_attention(gq, gk, gv, gu, go, T, 128 ** 0.5, part=gpart if ns > 1 else None, nsplit=ns)


def d32_x6(T, brows):
    ns = attn_splits(N, H, _planned_len(T, N, brows), wg_waves=4)
    aot_hip.attention_x6(q, xbank, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)


def gated_x6(T, brows):
    ns = gated_splits_x6(N, 1, _planned_len(T, N, brows))
    aot_hip.gated_attention_x6(gq, gbank, gu, go, T, 128 ** 0.5, part=gpart if ns > 1 else None, nsplit=ns)


n = 0
if mode == 'm14':
    xbank = aot_hip.x6_bank(1, 14 * N, C, 'cuda')
    aot_hip.attention_pack_x6(k, v, xbank, 14 * N)
    gbank = aot_hip.x6_gated_bank(1, 14 * N, 128, E, 'cuda')
    aot_hip.gated_pack_x6(gk, gv, gbank, 14 * N)
    for _ in range(3):
        d32(14 * N, CAP * N); gated(14 * N, CAP * N); d32_x6(14 * N, CAP * N); gated_x6(14 * N, CAP * N); n += 4
else:
    if mode == 'aotx6':                                    # the bf16x6 twin over the same launch mix: the bank appended frame by frame
        xbank = aot_hip.x6_bank(1, CAP * N, C, 'cuda')
        for slot in range(14):
            aot_hip.attention_pack_x6(k[slot * N:(slot + 1) * N], v[slot * N:(slot + 1) * N], xbank, N, slot=slot)
        sbank = aot_hip.x6_bank(1, N, C, 'cuda')
        aot_hip.attention_pack_x6(k[:N], v[:N], sbank, N)

        def fn(T, brows):
            ns = attn_splits(N, H, _planned_len(T, N, brows), wg_waves=4)
            aot_hip.attention_x6(q, sbank if brows == N else xbank, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)
    elif mode == 'gatedx6':                                # DeAOT's bf16x6 twin: packed K / V banks appended frame by frame
        gbank = aot_hip.x6_gated_bank(1, CAP * N, 128, E, 'cuda')
        for slot in range(14):
            aot_hip.gated_pack_x6(gk[slot * N:(slot + 1) * N], gv[slot * N:(slot + 1) * N], gbank, N, slot=slot)
        gsbank = aot_hip.x6_gated_bank(1, N, 128, E, 'cuda')
        aot_hip.gated_pack_x6(gk[:N], gv[:N], gsbank, N)

        def fn(T, brows):
            ns = gated_splits_x6(N, 1, _planned_len(T, N, brows))
            aot_hip.gated_attention_x6(gq, gsbank if brows == N else gbank, gu, go, T, 128 ** 0.5, part=gpart if ns > 1 else None, nsplit=ns)
    else:
        fn = d32 if mode == 'aot' else gated
    for t in range(1, 70):
        M = 1 + (t - 1) // 5
        for layer in range(3):
            fn(N, N); fn(M * N, CAP * N)       # self-attention over the frame, long-term attention over the bank
            n += 2
torch.cuda.synchronize()
print('launches', n)
