import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o, oracle_c as oc
oc.build()
sizes=[(2107, 2053), (2137, 2072), (2132, 2077), (2120, 2064)]
rng=np.random.default_rng(3)
inst=capi.Instance(0); solo=capi.Instance(0); solo.set_solver(method=capi.SC_METHOD_MULTIGRID)
for kind in ("smooth","noise"):
    items=[]
    for k,(W,H) in enumerate(sizes):
        dst,patch,mask,cx,cy=o.synth_inputs(W,H,seed_dst=k,seed_patch=10+k,margin=24)
        if kind=="noise":
            dst=rng.integers(0,256,dst.shape,dtype=np.uint8); patch=rng.integers(0,256,patch.shape,dtype=np.uint8)
        items.append((dst,patch,mask,cx,cy))
    jobs=capi.Pool.make_jobs(len(items)); keep=[]
    for j,(dst,patch,mask,cx,cy) in zip(jobs,items):
        f,b0,b,m=inst.to_device(patch),inst.to_device(dst),inst.to_device(dst),inst.to_device(mask)
        keep.append((f,b0,b,m,dst.shape))
        j.face,j.face_cols,j.face_rows,j.face_step=f,patch.shape[1],patch.shape[0],3*patch.shape[1]
        j.body,j.body_cols,j.body_rows,j.body_step=b,dst.shape[1],dst.shape[0],3*dst.shape[1]
        j.mask,j.mask_cols,j.mask_rows,j.mask_step=m,mask.shape[1],mask.shape[0],mask.shape[1]
        j.centerX,j.centerY,j.body_restore=cx,cy,b0
    rc=inst.run_device_batch(jobs); i=inst.info()
    print(kind,"rc",rc,[j.rc for j in jobs],"members",i.group_members,i.group_ragged,"cycles",i.sweeps,"last_update",i.last_update,flush=True)
    for k,((f,b0,b,m,shape),it) in enumerate(zip(keep,items)):
        got=inst.from_device(b,shape)
        want=oc.seamless_clone(it[0],it[1],it[2],it[3],it[4],nthreads=16,exact_den=False)
        d=np.abs(got.astype(np.int16)-want.astype(np.int16))
        body=it[0].copy(); solo.run(it[1],body,it[2],it[3],it[4])
        print("  member",k,"max vs port",int(d.max()),"pct %.3f"%(100*np.count_nonzero(d)/d.size),"solo cycles",solo.info().sweeps,"bytes differing from solo",int((body!=got).sum()),flush=True)
    for kp in keep:
        for p in kp[:4]: inst.free(p)
