# LDS bank-conflict model (MI355X_MICROARCH.md, LDS): ds_read_b128 in 4 groups of 16 lanes, 64 banks; ds_write_b64 in 4 groups of 16 contiguous lanes, 32 banks
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[l+32 for l in g] for g in G128]
def cost(groups, addr, width_dw, banks):
    tot=0
    for g in groups:
        per={}
        for l in g:
            a=addr(l)//4
            for d in range(width_dw):
                per.setdefault((a+d)%banks,set()).add(a+d)
        tot+=max(len(s) for s in per.values())
    return tot
def layout(pad, xor):
    POS=256+16*pad
    def off(index, chunk):
        sw=(index&15) if xor else 0
        return index*POS+((chunk^sw)*16)
    return off
S=21
for pad,xor in [(0,True),(1,False),(2,False),(3,False),(4,False),(5,False),(6,False)]:
    off=layout(pad,xor)
    res=[]
    for kc in range(4):
      for base in range(0,3):
        # column-tile read: lane (r=l&15,q4=l>>4) index = base + 21 r
        c=cost(G128, lambda l: off(1+S+base+S*(l&15), kc*4+(l>>4)), 4, 64)
        # tail-tile read: consecutive positions
        t=cost(G128, lambda l: off(1+S+336+base+(l&15), kc*4+(l>>4)), 4, 64)
        res.append((c,t))
    # stores: ds_write_b64, lane (r,q4): channel ch=(mg*2+i)*16+4*q4 -> chunk=ch//8, sub=(ch%8)*2 bytes
    W=[list(range(16*g,16*g+16)) for g in range(4)]
    ws=[]
    for mi in range(8):
        wc=cost(W, lambda l: off(1+S+S*(l&15)+3, (mi*16+4*(l>>4))//8)+((mi*16+4*(l>>4))%8)*2, 2, 32)
        wt=cost(W, lambda l: off(1+S+336+(l&15), (mi*16+4*(l>>4))//8)+((mi*16+4*(l>>4))%8)*2, 2, 32)
        ws.append((wc,wt))
    print("pad",pad,"xor",xor,"read col/tail cycles (ideal 4):",sorted(set(res)),"write col/tail (ideal 4):",sorted(set(ws)), "plane KB", 477*(256+16*pad)/1024)
