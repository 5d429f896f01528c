import re,sys
sys.path.insert(0,'/root/repo/tools/isa')
from blocks import classify
def loops(path, key="queue_kernelILb1ELi2ELb1ELb0E"):
    txt=open(path).read().split("\n")
    start=[i for i,l in enumerate(txt) if re.match(r"^_Z\S*"+key+r"\S*:",l)][0]
    end=[i for i in range(start,len(txt)) if txt[i].strip().startswith("s_endpgm")][0]
    cur=("none",0); res={}
    for i in range(start,end):
        l=txt[i]
        m=re.match(r"^(\.LBB\S+):\s*;\s*(.*)$",l)
        if m:
            ann=m.group(2)
            h=re.search(r"Header=(BB\d+_\d+) Depth=(\d+)",ann)
            p=re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)",ann)
            if h: cur=(h.group(1),int(h.group(2)))
            elif p: cur=(m.group(1).strip(".L"), int(p.group(2))+1)
            elif "Loop Header" in ann: cur=(m.group(1).strip(".L"),1)
            else: cur=("none",0)
            continue
        if re.match(r"^\.LBB\S+:",l): cur=("none",0); continue
        t=l.strip()
        if not t or t.startswith((";",".","//")): continue
        c=classify(t.split()[0])
        res.setdefault(cur,{}).setdefault(c,0); res[cur][c]+=1
    m=re.search(key+r".*?\.sgpr_spill_count: (\d+).*?\.vgpr_spill_count: (\d+)", open(path).read(), re.S)
    return res
KEY={"f32":"queue_kernelILb1ELi2ELb1ELb0E","f32x":"queue_kernelILb1ELi3ELb1ELb0E","f64":"queue_kernelILb1ELi1ELb1ELb0E","none":"queue_kernelILb1ELi0ELb1ELb0E"}
r=loops(sys.argv[1], KEY.get(sys.argv[2] if len(sys.argv)>2 else "f32x", sys.argv[2] if len(sys.argv)>2 else ""))
# pass-1 loops: the two depth-3 loops with the most v64 (body CORR=false first in file order -> smaller label number)
d3=[(k,v) for k,v in r.items() if k[1]==3 and v.get("v64",0)>150]
d3.sort(key=lambda kv:int(kv[0][0].split("_")[1]))
out=[]
for k,v in d3: out.append(f"p1[{k[0]}] v64={v.get('v64',0)} v={v.get('v',0)} lane={v.get('lane',0)} s={v.get('s',0)}")
p2=[(k,v) for k,v in r.items() if k[1]==3 and v.get("trans",0)>=5]
p2.sort(key=lambda kv:int(kv[0][0].split("_")[1]))
for k,v in p2: out.append(f"p2[{k[0]}] v64={v.get('v64',0)} v={v.get('v',0)} lane={v.get('lane',0)} s={v.get('s',0)}")
tot=sum(v.get('lane',0) for k,v in r.items() if k[1]>=2)
print(sys.argv[1].split('/')[-1], " | ".join(out), "| lane(depth>=2)=",tot)
