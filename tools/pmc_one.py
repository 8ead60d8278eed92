import csv, sys, json
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'mlp_forward16_kernel' in r['Kernel_Name']]
best=max(rows, key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print(json.dumps({'counter':best['Counter_Name'],'value_KB':float(best['Counter_Value']),'ms':(int(best['End_Timestamp'])-int(best['Start_Timestamp']))/1e6,'grid':best['Grid_Size']}))
