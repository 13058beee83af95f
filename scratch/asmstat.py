import re,collections,sys
pat=sys.argv[1]
txt=open('/tmp/kernels.s').read().split('\n')
bounds=[(i,re.match(r'([_A-Za-z0-9]+):\s*; @',l).group(1)) for i,l in enumerate(txt) if re.match(r'([_A-Za-z0-9]+):\s*; @',l)]
for bi,(i,name) in enumerate(bounds):
    if pat not in name: continue
    end=bounds[bi+1][0] if bi+1<len(bounds) else len(txt)
    ops=collections.Counter(re.findall(r'^\s+([a-z_0-9]+)\s', '\n'.join(txt[i:end]), flags=re.M))
    print(name[:100], sum(ops.values()))
    for k,v in ops.most_common(300):
        if any(t in k for t in ('flat','global','ds_','scratch','buffer','s_load','accvgpr','saveexec','s_cbranch','s_waitcnt','swappc','fma_f64','fmac_f64')): print('  ',k,v)
