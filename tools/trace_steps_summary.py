import sys,re,ast
for line in sys.stdin:
    if line.startswith("step ms:"):
        v=ast.literal_eval(line[len("step ms:"):].strip())
        import statistics
        big=[(i,x) for i,x in enumerate(v) if x>2*statistics.median(v)]
        print("n",len(v),"median",statistics.median(v),"mean",round(sum(v)/len(v),3),"outliers",big[:10])
        # windows
        w=50
        print("window means",[round(sum(v[i:i+w])/w,3) for i in range(0,len(v)-w+1,w)])
