"""LDS bank model of wgrad3w_f16_kernel's two images (csrc/se_wgrad.hip): cycles per 64-row step of a workgroup under the round-3 cell
maps and the round-6 ones.  The lane groups and bank widths are those of MI355X_MICROARCH.md, section LDS (lds_bank_model.py).
usage: python tools/micro/lds_wgrad3w_model.py"""
from lds_bank_model import cyc_b128, cyc_b32, cyc_w64


def step_cycles(ycell, xcell, PLY, PLX, one_dword_read):
    """(read cycles, write cycles) of one 64-row step: 4 waves x 4 k-steps x 2 planes; ycell(r, ch) / xcell(r) in 16-byte cells"""
    tr = tw = 0
    for wave in range(4):
        for ks in range(4):
            for pl in range(2):
                for a in range(2):      # dY fragments: row a * 32 + (lane & 31), cell (lane >> 5) + 2 ks
                    tr += cyc_b128([ycell(a * 32 + (l & 31), (l >> 5) + 2 * ks) * 16 + pl * PLY for l in range(64)])
                cen = [(xcell(wave * 32 + (l & 31)) + 2 * ks + (l >> 5) + 1) * 16 + pl * PLX for l in range(64)]
                tr += cyc_b128(cen)
                if one_dword_read:      # kg = 0: last dword of cell 2 ks; kg = 1: first dword of cell 2 ks + 3 (the inner ones: v_permlane32_swap)
                    tr += cyc_b32([(xcell(wave * 32 + (l & 31)) + 2 * ks + (3 if l >> 5 else 0)) * 16 + (0 if l >> 5 else 12) + pl * PLX for l in range(64)])
                else:
                    tr += cyc_b32([c - 4 for c in cen]) + cyc_b32([c + 16 for c in cen])
        for r0, img in ((0, 'x'), (64, 'x'), (0, 'y')):      # transposing 8-byte stores: lane (q = l & 15, rg = 4 wave + (l >> 4)), row r0 + 4 q + j
            for j in range(4):
                for pl in range(2):
                    ad = []
                    for l in range(64):
                        q, rg = l & 15, wave * 4 + (l >> 4)
                        r = r0 + 4 * q + j
                        ad.append(xcell(r) * 16 + 16 + 8 * rg + pl * PLX if img == 'x' else ycell(r, rg >> 1) * 16 + 8 * (rg & 1) + pl * PLY)
                    tw += cyc_w64(ad)
    return tr, tw


def gy(r):
    return (((r >> 4) & 1) << 2) | (((r >> 3) & 1) << 1) | (((r >> 1) ^ (r >> 2)) & 1)


if __name__ == '__main__':
    old = step_cycles(lambda r, ch: 9 * r + (r >> 4) + ch, lambda r: 10 * r + (r >> 3), (9 * 64 + 4) * 16, (10 * 128 + 16) * 16, False)
    new = step_cycles(lambda r, ch: 8 * r + (ch ^ gy(r)), lambda r: 95 * (r >> 3) + 12 * (r & 7), 64 * 8 * 16, 1519 * 16, True)
    ideal_r, ideal_w = 4 * 4 * 2 * (2 * 4 + 4 + 2), 4 * 3 * 4 * 2 * 4
    for name, (r, w) in (('round 3 maps, two dword reads', old), ('round 6 maps, one dword read + lane exchange', new)):
        print(f'{name:48s} reads {r:5d}  writes {w:4d}  total {r + w:5d} LDS cycles per 64-row step (conflict-free: reads {ideal_r}, writes {ideal_w})')
