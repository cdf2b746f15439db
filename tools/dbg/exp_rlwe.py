"""development aid: time rlwe encrypt with experimental builds of the library (libsafecrypto_amd/lib/exp_*/libsc_hip.so); results of
the exp builds are not checked (they are wrong on purpose: pieces of the kernel are switched off to see what they cost)"""
import sys, os, json, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)


def one(lib, tag, batch):
    import torch
    import libsafecrypto_amd.engine as E
    if lib != 'default':
        E.load_library(os.path.join(ROOT, 'libsafecrypto_amd', 'lib', lib, 'libsc_hip.so'))
    from tests.golden_io import load_golden
    G = load_golden('rlwe_ref.txt')
    eng = E.Engine(0)
    r = eng.rlwe(G[f'{tag}_params'])
    r.set_arith(int(os.environ.get('RL_ARITH', '0')))
    r.set_pubkey(G[f'{tag}_kg_a0'], G[f'{tag}_kg_p0']); r.set_privkey(G[f'{tag}_kg_r2_0'])
    g = torch.Generator(device='cuda'); g.manual_seed(17)
    msgs = torch.randint(0, 256, (batch, r.msg_bytes), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, r.enc_words), dtype=torch.int32, device='cuda', generator=g)
    r.encrypt(msgs, tapes); torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ct = r.encrypt(msgs, tapes); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    ok = None
    if lib == 'default' or lib.endswith('_ok'):
        from tests import oracle_lib as O
        R = O.Rlwe(G[f'{tag}_params'])
        ok = all(R.encrypt(G[f'{tag}_kg_a0'], G[f'{tag}_kg_p0'], bytes(msgs[i].cpu().numpy()), tapes[i].cpu().numpy().view('uint32'))[1] == bytes(ct[i].cpu().numpy())
                 for i in (0, 1, batch // 2, batch - 1)) if hasattr(R, 'encrypt') else None
    print(json.dumps({'lib': lib, 'set': tag, 'batch': batch, 'us': round(best * 1e6, 1), 'Mops': round(batch / best / 1e6, 1),
                      'frac_hbm': round(batch * (r.enc_words * 4 + r.msg_bytes + ct.shape[1]) / best / 8e12, 3), 'parity': ok}), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2:
        one(sys.argv[1], sys.argv[2], int(sys.argv[3]))
    else:
        libs = sys.argv[1].split(',')
        for lib in libs:
            for tag in ('rlwe0', 'rlwe1'):
                subprocess.run([sys.executable, __file__, lib, tag, '262144'])
