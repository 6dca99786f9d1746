import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*memory_copy_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
big=[r for r in rows if int(r.get('Bytes', r.get('Size','0')) or 0) > 1<<20] if rows and ('Bytes' in rows[0] or 'Size' in rows[0]) else rows
print(rows[0].keys())
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
d.sort()
print("copies",len(d),"median us",d[len(d)//2],"p90",d[int(len(d)*0.9)],"max",d[-1])
long=[x for x in d if x>100]
print("long copies (>100us):",len(long),"median",long[len(long)//2] if long else None, "min",long[0] if long else None,"max",long[-1] if long else None)
