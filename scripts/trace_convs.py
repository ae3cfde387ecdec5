import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
nsteps=int(sys.argv[2])
convs=[r for r in rows if 'conv_fwd' in r['Kernel_Name'] or 'conv_wgrad' in r['Kernel_Name']]
per=len(convs)//nsteps
last=convs[-per:]
out=[]
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    k=r['Kernel_Name']
    nm=('fwd128' if '128' in k else 'fwd64') if 'conv_fwd' in k else 'wgrad'
    out.append("%s/%s/%s:%.0f"%(nm, r['Grid_Size_X'], r['LDS_Block_Size'], d))
print(" ".join(out))
