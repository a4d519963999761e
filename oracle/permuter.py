"""numpy restatement of DualGrainSeperatePermuter (reference modules/dynamic_modules/permuter.py:7-135).

TEST INFRASTRUCTURE ONLY (see oracle/dvq_oracle.c header).  Pinned by the reference's own
known-answer self-test (permuter.py:139-307, captured in tests/golden/permuter_*.npz by
oracle/gen_golden_permuter.py) and by synthetic cases run through the imported reference.
"""
import numpy as np


def forward(indices, grain, coarse_hw=16, fine_hw=32, content_pad=1024, content_eos=1025,
            cpos_pad=256, cpos_eos=257, fpos_pad=1024, fpos_eos=1025, order="region-first"):
    """permuter.py:50-109.  indices [B, fine_hw, fine_hw], grain [B, coarse_hw, coarse_hw] (0 coarse / 1 fine)."""
    indices = np.asarray(indices, np.int64)
    grain = np.asarray(grain, np.int64)
    B = indices.shape[0]
    hw1, hw2 = coarse_hw, fine_hw // coarse_hw
    # "B (h1 h2) (w1 w2) -> B h1 w1 (h2 w2)"
    cells = indices.reshape(B, hw1, hw2, hw1, hw2).transpose(0, 1, 3, 2, 4).reshape(B, hw1, hw1, hw2 * hw2)
    pos_fine = np.arange(fine_hw * fine_hw, dtype=np.int64).reshape(fine_hw, fine_hw)
    pos_fine_region = pos_fine.reshape(hw1, hw2, hw1, hw2).transpose(0, 2, 1, 3).reshape(hw1, hw1, hw2 * hw2)
    pos_coarse = np.arange(hw1 * hw1, dtype=np.int64)
    cc, cp, fc, fp = [], [], [], []
    for i in range(B):
        g = grain[i]
        cc.append(np.concatenate([cells[i][:, :, 0][g == 0], [content_eos]]))            # :60-61
        cp.append(np.concatenate([pos_coarse[g.reshape(-1) == 0], [cpos_eos]]))         # :72
        if order == "region-first":
            fc.append(np.concatenate([cells[i][g == 1].reshape(-1), [content_eos]]))    # :80
            fp.append(np.concatenate([pos_fine_region[g == 1].reshape(-1), [fpos_eos]]))  # :84
        elif order == "row-first":
            gf = g.repeat(hw2, axis=-1).repeat(hw2, axis=-2)                             # :88 (hw2 = 2)
            fc.append(np.concatenate([indices[i][gf == 1].reshape(-1), [content_eos]]))  # :89
            fp.append(np.concatenate([pos_fine[gf == 1], [fpos_eos]]))                   # :93
        else:
            raise NotImplementedError(order)

    def pad(seqs, value):                                                                # pad_sequence
        L = max(len(s) for s in seqs)
        out = np.full((B, L), value, np.int64)
        for i, s in enumerate(seqs):
            out[i, :len(s)] = s
        return out

    coarse_content, fine_content = pad(cc, content_pad), pad(fc, content_pad)
    return dict(coarse_content=coarse_content, fine_content=fine_content,
                coarse_position=pad(cp, cpos_pad), fine_position=pad(fp, fpos_pad),
                coarse_segment=np.zeros_like(coarse_content), fine_segment=np.ones_like(fine_content))


def forward_back(coarse_content, fine_content, coarse_position, fine_position,
                 coarse_hw=16, fine_hw=32, cpos_eos=257, fpos_eos=1025):
    """permuter.py:111-135 (sequential: a later entry overwrites an earlier one at the same position)."""
    B, Lc = coarse_content.shape
    Lf = fine_content.shape[1]
    hw1, hw2 = coarse_hw, fine_hw // coarse_hw
    tc = np.zeros((B, hw1 * hw1), np.int64)
    tgt = np.zeros((B, fine_hw * fine_hw), np.int64)
    for i in range(B):
        for k in range(Lc):
            if coarse_position[i, k] == cpos_eos:
                up = tc[i].repeat(hw2 * hw2)                                   # (h1 w1 h2 w2)
                tgt[i] = up.reshape(hw1, hw1, hw2, hw2).transpose(0, 2, 1, 3).reshape(-1)   # -> (h1 h2 w1 w2)
                break
            tc[i, coarse_position[i, k]] = coarse_content[i, k]
        for k in range(Lf):
            if fine_position[i, k] == fpos_eos:
                break
            tgt[i, fine_position[i, k]] = fine_content[i, k]
    return tgt.reshape(B, fine_hw, fine_hw)
