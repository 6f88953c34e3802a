import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
t0=min(int(r['Start_Timestamp']) for r in rows)
lz=[(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0) for r in rows if 'lz4f_compress' in r['Kernel_Name']]
print('lz4 kernels (ms):',[(round(a/1e6),round(b/1e6)) for a,b in lz])
tiles=[(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0) for r in rows if r['Kernel_Name'].startswith('rh_tile')]
print('tile launches',len(tiles))
# frames per 100 ms bucket
import collections
b=collections.Counter(int(a/1e8) for a,_ in tiles)
print('tile launches per 100ms:',[b[i] for i in range(max(b)+1)])
d=[e-s for s,e in tiles]
print('tile dur avg us',sum(d)/len(d)/1e3)
for a,bb in lz:
    inside=[e-s for s,e in tiles if s>=a and e<=bb]
    if inside: print('during lz4',round(a/1e6),': n',len(inside),'avg us',sum(inside)/len(inside)/1e3)
names=collections.Counter()
for r in rows: names[r['Kernel_Name'][:40]]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
print({k:round(v/1e6) for k,v in names.most_common(8)})
mc=glob.glob(sys.argv[1]+'/*/*memory_copy_trace.csv')
if mc:
    cr=list(csv.DictReader(open(mc[0])))
    big=[(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0,r.get('Direction','')) for r in cr if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>2e5]
    b2=collections.Counter(int(a/1e8) for a,_,_ in big)
    print('long copies (>0.2 ms) per 100ms:',[b2[i] for i in range(max(b2)+1)] if b2 else [])
    tot=collections.Counter()
    for a,b,d in big: tot[d]+=b-a
    print({k:round(v/1e6) for k,v in tot.items()})
