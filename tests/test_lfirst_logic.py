"""The flag logic of k_lf_reduce (dark_amd/csrc/lfirst.inc), restated lane by lane in Python and checked against its definition by brute force.

A tile of 2048 slots, 256 threads x 8 slots: heads (the key changes) and reasons (two symbols in front inside a group, or suffix 0).  The kernel
decides every group that lies inside the tile with two segmented ORs over the waves' ballots and marks the stretches of groups that cross the
tile's borders (LF_EDGE_FIRST / LF_EDGE_LAST) for k_lf_straddle.  `sim` follows the kernel's arithmetic (ballots as Python integers, the same
per-wave summaries, the same two scans); `brute` is the definition: a group inside the tile is live iff it has more than one member and a reason.
This is the model the kernel was written from -- it ran (3000 random tiles) before the kernel ever did; the kernel itself is tested on the GPU
(tests/test_gpu_parity.py::test_lfirst_bwt_without_the_suffix_array, the fuzz campaign with DK_LFIRST=2)."""
import random
IPT=8; WAVES=4; BLOCK=256; TILE=2048
def ctz(x): return (x & -x).bit_length()-1
def clz64(x): return 64 - x.bit_length()
def sim(head, nu, count_in_tile, next_is_head, last_tile):
    # head[a], nu[a] for a in tile (real slots a<count_in_tile); virtual heads beyond
    flags=[0]*TILE
    H=[0]*BLOCK; NU=[0]*BLOCK; valid=[0]*BLOCK; nexth=[True]*BLOCK
    for t in range(BLOCK):
        for j in range(IPT+1):
            a=t*IPT+j
            if a<TILE:
                hd = head[a] if a<count_in_tile else True
            else:
                hd = next_is_head if count_in_tile==TILE else True
            if j==IPT: nexth[t]=hd; break
            if a<count_in_tile: valid[t]|=1<<j
            if hd: H[t]|=1<<j
            if a<count_in_tile and nu[a]: NU[t]|=1<<j
    s_w=[[0]*5 for _ in range(WAVES)]
    closed = nexth[BLOCK-1]
    per=[]
    for w in range(WAVES):
        BH=BA=BS=BP=0
        info=[]
        for l in range(64):
            t=w*64+l
            hasH=H[t]!=0
            first=ctz(H[t]) if hasH else IPT
            last=H[t].bit_length()-1 if hasH else -1
            pre=(NU[t]&((1<<first)-1))!=0
            suf=((NU[t]>>last)!=0) if hasH else NU[t]!=0
            if hasH: BH|=1<<l
            if NU[t]: BA|=1<<l
            if suf: BS|=1<<l
            if pre: BP|=1<<l
            info.append((hasH,first,last,pre,suf))
        wh=BH!=0
        pl=BH.bit_length()-1 if wh else 0; pf=ctz(BH) if wh else 0
        s_w[w][0]=1 if wh else 0
        s_w[w][1]= (1 if (((BS>>pl)&1) or (pl<63 and (BA>>(pl+1))!=0)) else 0) if wh else (1 if BA else 0)
        s_w[w][2]= (1 if (((BP>>pf)&1) or (BA & ((1<<pf)-1))) else 0) if wh else (1 if BA else 0)
        per.append((BH,BA,BS,BP,info))
    ns_tot=nh_tot=0
    for w in range(WAVES):
        BH,BA,BS,BP,info=per[w]
        for l in range(64):
            t=w*64+l
            hasH,first,last,pre_or,suf_or=info[l]
            lt=(1<<l)-1; gt=0 if l==63 else (~((2<<l)-1)) & ((1<<64)-1)
            hb=BH&lt
            if hb:
                p=hb.bit_length()-1
                fwd=((BS>>p)&1)!=0 or (BA & lt & ~((2<<p)-1))!=0; head_before=True
            else:
                fwd=(BA&lt)!=0; head_before=False
                v=w-1
                while v>=0 and not head_before:
                    fwd=fwd or s_w[v][1]!=0; head_before=s_w[v][0]!=0; v-=1
            ha=BH&gt
            if ha:
                q=ctz(ha)
                bwd=((BP>>q)&1)!=0 or (BA & gt & ((1<<q)-1))!=0; head_after=True
            else:
                bwd=(BA&gt)!=0; head_after=False
                v=w+1
                while v<WAVES and not head_after:
                    bwd=bwd or s_w[v][2]!=0; head_after=s_w[v][0]!=0; v+=1
                if not head_after and (last_tile or closed): head_after=True
            j=0
            if first>0:
                goes_on=not hasH
                reason=fwd or pre_or or (goes_on and bwd)
                ef=not head_before; el=goes_on and not head_after
                f=((8 if ef else 0)|(16 if el else 0)) if (ef or el) else (2 if reason else 0)
                while j<first:
                    if not (valid[t]>>j)&1: break
                    flags[t*IPT+j]=f; ns_tot+= 1 if f&2 else 0; j+=1
                j=first
            while j<IPT and (valid[t]>>j)&1:
                rest=H[t]>>(j+1)
                nxt=j+1+ctz(rest) if rest else IPT
                open_end = nxt==IPT and not nexth[t]
                reason=((NU[t]>>j)&((1<<(nxt-j))-1))!=0 or (open_end and bwd)
                single= nxt==j+1 and (nxt<IPT or nexth[t])
                el=open_end and not head_after
                f=16 if el else (2 if (reason and not single) else 0)
                nh_tot += 1 if f&2 else 0
                for q in range(j,nxt):
                    if not (valid[t]>>q)&1: break
                    flags[t*IPT+q]=f|(1 if q==j else 0); ns_tot+=1 if f&2 else 0
                j=nxt
    return flags, ns_tot, nh_tot
def brute(head, nu, cnt, next_is_head, last_tile):
    flags=[0]*TILE
    a=0
    # group boundaries in tile: group starting before tile iff head[0] False
    starts=[i for i in range(cnt) if head[i]]
    bounds=[]
    if not head[0]: bounds.append((None,0))
    # build groups
    i=0
    groups=[]
    cur_start=None if not head[0] else 0
    seg_begin=0
    for i in range(1,cnt+1):
        if i==cnt or head[i]:
            groups.append((cur_start, seg_begin, i))
            cur_start=i; seg_begin=i
    ns=nh=0
    for (st, b, e) in groups:
        edge_first = st is None
        closes = True
        if e==cnt:
            if cnt==TILE and not last_tile: closes = next_is_head
            else: closes=True
        edge_last = not closes
        reason=any(nu[b:e])
        for x in range(b,e):
            if edge_first or edge_last:
                f=(8 if edge_first else 0)|(16 if edge_last else 0)
            else:
                f=2 if (reason and (e-b)>1) else 0
            if x==b and st is not None: f|=1
            flags[x]=f
            if f&2: ns+=1
        if not(edge_first or edge_last) and reason and (e-b)>1: nh+=1
    return flags,ns,nh


def test_tile_flags_match_the_definition():
    random.seed(1)
    for trial in range(400):
        cnt = TILE if random.random() < 0.7 else random.randint(1, TILE)
        ph = random.choice([0.001, 0.01, 0.1, 0.5, 0.9])
        pn = random.choice([0.0, 0.001, 0.01, 0.2])
        head = [random.random() < ph for _ in range(TILE)]
        if random.random() < 0.5:
            head[0] = True
        nu = [(random.random() < pn) for _ in range(TILE)]
        nih = random.random() < 0.5
        lt = (cnt < TILE) or random.random() < 0.2
        f1, ns1, nh1 = sim(head, nu, cnt, nih, lt)
        f2, ns2, nh2 = brute(head, nu, cnt, nih, lt)
        assert f1 == f2 and ns1 == ns2 and nh1 == nh2, (trial, cnt, ph, pn, nih, lt)


def _pivot_order(t, members, depth):
    """k_lf_deep's rule (lfirst.inc, step 1): measure every member against the first one (the pivot) -- bytes in common behind `depth`, and who
    is smaller where they part (the end of the text: the shorter suffix is smaller); smaller ones first by c ascending, then the pivot, then
    the larger ones by c descending; members of one side with one c go on as a subgroup at depth + c."""
    if len(members) <= 1:
        return list(members)
    n = len(t)
    piv = members[0]
    classes = {}
    for m in members[1:]:
        c = 0
        while piv + depth + c < n and m + depth + c < n and t[piv + depth + c] == t[m + depth + c]:
            c += 1
        po, pr = m + depth + c, piv + depth + c
        smaller = True if po >= n else False if pr >= n else t[po] < t[pr]
        classes.setdefault((0, c) if smaller else (2, -c), []).append(m)
    out = []
    for key in sorted(k for k in classes if k[0] == 0):
        out += _pivot_order(t, classes[key], depth + key[1])
    out.append(piv)
    for key in sorted(k for k in classes if k[0] == 2):
        out += _pivot_order(t, classes[key], depth - key[1])
    return out


def test_pivot_order_is_the_suffix_order():
    random.seed(7)
    for trial in range(60):
        kind = trial % 4
        if kind == 0:
            t = bytes(random.choice(b"ab") for _ in range(300))
        elif kind == 1:
            t = b"x" * random.randint(50, 200) + bytes(random.choice(b"xyz") for _ in range(100)) + b"x" * random.randint(10, 90)
        elif kind == 2:
            t = (b"abc" * 70)[: random.randint(100, 210)] + b"q" + b"abc" * 20
        else:
            seg = bytes(random.randrange(4) for _ in range(60))
            t = seg + bytes([9]) + seg + bytes([7]) + seg[:30] + b"\0\0\0"
        n = len(t)
        # a group = the suffixes that share their first symbol (depth 1), as after a sort on one symbol
        for first in set(t):
            members = [i for i in range(n) if t[i] == first]
            random.shuffle(members)
            got = _pivot_order(t, members, 1)
            want = sorted(members, key=lambda i: t[i:])  # bytes comparison: a proper prefix is smaller (src/saca.rs:105-113)
            assert got == want, (trial, first)
