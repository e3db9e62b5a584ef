R = [[0,36,3,41,18],[1,44,10,45,2],[62,6,43,15,61],[28,55,25,21,56],[27,20,39,8,14]]  # R[x][y]
def rnd(src, dst, rc):
    out=[]
    # D from C (C of src assumed in c0..c4)
    for x in range(5):
        out.append(f"d{x} = c{(x+4)%5} ^ rotl64(c{(x+1)%5}, 1);")
    first=True
    for Y in range(5):
        for X in range(5):
            x=(X+3*Y)%5; y=X
            r=R[x][y]
            e=f"({src}{x+5*y} ^ d{x})"
            out.append(f"b{X} = " + (f"rotl64({e}, {r});" if r else f"{e};"))
        for X in range(5):
            s=f"{dst}{X+5*Y} = b{X} ^ (~b{(X+1)%5} & b{(X+2)%5});"
            out.append(s)
        if Y==0:
            out.append(f"{dst}0 ^= {rc};")
        for X in range(5):
            out.append(f"n{X} {'=' if Y==0 else '^='} {dst}{X+5*Y};")
    for X in range(5):
        out.append(f"c{X} = n{X};")
    return "\n        ".join(out)
print("inline void keccak_f1600_fast(uint64_t s[25]) {")
print("    uint64_t " + ", ".join(f"a{i} = s[{i}]" for i in range(25)) + ";")
print("    uint64_t " + ", ".join(f"e{i}" for i in range(25)) + ";")
print("    uint64_t b0, b1, b2, b3, b4, d0, d1, d2, d3, d4, n0, n1, n2, n3, n4;")
print("    uint64_t " + ", ".join(f"c{x} = a{x} ^ a{x+5} ^ a{x+10} ^ a{x+15} ^ a{x+20}" for x in range(5)) + ";")
print("    for (int round = 0; round < 24; round += 2) {")
print("        " + rnd("a","e","KECCAK_RC[round]"))
print("        " + rnd("e","a","KECCAK_RC[round + 1]"))
print("    }")
print("    " + " ".join(f"s[{i}] = a{i};" for i in range(25)))
print("}")
