import itertools
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[l+32 for l in g] for g in G128]
def cyc_b128(addrs):   # addrs per lane (bytes, 16B aligned) -> total LDS cycles (4 ideal)
    tot=0
    for g in G128:
        banks={}
        for l in g:
            a=addrs[l]
            for b in range(4):
                banks.setdefault(((a//4)+b)%64,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
def cyc_b32(addrs):    # 2 ideal
    tot=0
    for h in (range(32),range(32,64)):
        banks={}
        for l in h:
            a=addrs[l]; banks.setdefault((a//4)%32,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
def cyc_w64(addrs):    # 4 groups of 16 contiguous lanes, bank mod 32, 4 ideal
    tot=0
    for g0 in range(0,64,16):
        banks={}
        for l in range(g0,g0+16):
            a=addrs[l]
            for b in range(2): banks.setdefault(((a//4)+b)%32,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
def cyc_b64(addrs):    # ds_read_b64: 2 x 32, bank mod 64, 2 ideal
    tot=0
    for h in (range(32),range(32,64)):
        banks={}
        for l in h:
            a=addrs[l]
            for b in range(2): banks.setdefault(((a//4)+b)%64,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
