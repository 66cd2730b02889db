"""Per-kernel register / LDS / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage output (dev tool):
   hipcc ... -c conv.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [regex]"""
import sys,re,subprocess
txt=open(sys.argv[1]).read()
blocks=re.split(r'remark: [^\n]*Function Name: ',txt)
names=[b.split('\n')[0].strip() for b in blocks[1:]]
dem=subprocess.run(['/usr/bin/c++filt'],input='\n'.join(names),capture_output=True,text=True).stdout.split('\n')
pat=sys.argv[2] if len(sys.argv)>2 else '.'
for b,n in zip(blocks[1:],dem):
    def g(k):
        m=re.search(k+r': (\d+)',b); return int(m.group(1)) if m else -1
    n=re.sub(r'\(.*','',n.replace('void (anonymous namespace)::','').replace('(anonymous namespace)::',''))
    if re.search(pat,n):
        print('%-62s vgpr %3d agpr %3d scratch %3d occ %d lds %6d'%(n,g('VGPRs'),g('AGPRs'),g(r'ScratchSize \[bytes/lane\]'),g(r'Occupancy \[waves/SIMD\]'),g(r'LDS Size \[bytes/block\]')))
