import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
tiles=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('rh_tile_1')]
a=tiles[-4]; b=tiles[-3]
prev=None
for r in rows[a:b+1]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1e3:10.1f} dur {(e-s)/1e3:7.1f} gap {0 if prev is None else (s-prev)/1e3:6.1f} {r['Kernel_Name'][:34]}")
    prev=e
