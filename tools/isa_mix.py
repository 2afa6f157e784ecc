import re,sys,collections
FAST={'v_mul_f32','v_add_f32','v_sub_f32','v_subrev_f32','v_fma_f32','v_fmac_f32','v_mac_f32','v_mov_b32','v_and_b32','v_or_b32','v_xor_b32','v_ashrrev_i32','v_lshlrev_b32','v_lshrrev_b32','v_add_u32','v_sub_u32','v_subrev_u32','v_not_b32','v_accvgpr_write_b32','v_accvgpr_read_b32'}
s=open(sys.argv[1]).read(); pat=sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S|re.M):
    name,body=m.group(1),m.group(2)
    if pat not in name: continue
    blocks=[];cur=[];lab='entry'
    for l in body.splitlines():
        l=l.strip()
        if not l or l.startswith(';'): continue
        if re.match(r'\.LBB\d+_\d+:',l): blocks.append((lab,cur));cur=[];lab=l.split(':')[0]
        elif not l.startswith('.'): cur.append(l.split(';')[0].strip())
    blocks.append((lab,cur))
    print(name[:90])
    for lab,b in blocks:
        if len(b)<60: continue
        cnt=collections.Counter(re.sub(r'_(e32|e64|dpp|sdwa)$','',l.split()[0]) for l in b)
        v={k:n for k,n in cnt.items() if k.startswith('v_')}
        fast=sum(n for k,n in v.items() if k in FAST); slow=sum(v.values())-fast
        mem={k:n for k,n in cnt.items() if k.startswith(('ds_','global_','buffer_','flat_','s_barrier'))}
        print(f'  {lab}: {len(b)} instr, VALU {fast+slow} (fast {fast}, slow {slow}) est {fast*1.05+slow*1.85:.0f} ns-units;', dict(sorted(mem.items())))
        print('     slow:', {k:n for k,n in sorted(v.items(), key=lambda x:-x[1]) if k not in FAST})
