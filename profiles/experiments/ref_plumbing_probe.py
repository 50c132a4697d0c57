#!/usr/bin/env python3
"""REF-0 of BASELINE.md: plumbing-only ceiling of the REFERENCE's own transport.

Runs, in the build container only (the reference never travels to the GPU box), the reference's
`server.serve` (server.py:210-224) in its own process with the four multiprocessing.Queues of
`Soccer2DEnv` (soccer_2d_env.py:62-65), and drives it with the exact queue pattern of
`Soccer2DEnv.step` (soccer_2d_env.py:238-252): put one PlayerAction + one TrainerAction, wait for one
player State and one trainer State.  The two agents are fake in-process gRPC clients
(`GameStub.GetPlayerActions / GetTrainerActions`) sending a near-empty State: no rcssserver, no
proxy -- an UPPER bound of the reference chain's throughput.  The full chain is not measurable
offline (binaries are fetched from GitHub releases, scripts/download-*.sh).

    python profiles/experiments/ref_plumbing_probe.py [steps]   ->  profiles/r01/ref_plumbing_ceiling.json
"""
import json
import os
import sys
import tempfile
import threading
import time
from multiprocessing import Lock, Manager, Process, Queue

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get('S2D_REFERENCE', '/root/reference')
PORT = int(os.environ.get('S2D_PROBE_PORT', '50651'))


def _server(port, psq, tsq, taq, paq, log_dir):
    # as Soccer2DEnv._run_grpc does (soccer_2d_env.py:413-425): Manager / Lock live in the server process
    os.setsid()                              # own process group: the probe ends it (and the Manager's helper) as a group
    manager = Manager()
    lock = Lock()
    nconn = manager.Value('i', 0)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import _standins
    _standins.install()                      # server.py:12 imports pyrusgeom (unused)
    sys.path.insert(0, REF)
    from server import serve
    serve(port, lock, nconn, psq, tsq, taq, paq, log_dir)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: this probe only runs in the build container')
    tmp = tempfile.mkdtemp(prefix='s2d_ref_probe_')
    os.chdir(tmp)
    psq, tsq, taq, paq = Queue(), Queue(), Queue(), Queue()
    proc = Process(target=_server, args=(PORT, psq, tsq, taq, paq, tmp))
    proc.start()
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import _standins
    _standins.install()
    sys.path.insert(0, REF)
    import grpc
    import service_pb2 as pb2
    import service_pb2_grpc as pb2_grpc
    chan = grpc.insecure_channel(f'localhost:{PORT}')
    grpc.channel_ready_future(chan).result(timeout=30)
    stub = pb2_grpc.GameStub(chan)
    rr_p = stub.Register(pb2.RegisterRequest(agent_type=pb2.AgentType.PlayerT, team_name='probe', uniform_number=1, rpc_version=1))
    rr_t = stub.Register(pb2.RegisterRequest(agent_type=pb2.AgentType.TrainerT, team_name='probe', uniform_number=0, rpc_version=1))
    stop = threading.Event()

    def agent(rr, call):
        c = 0
        while not stop.is_set():
            s = pb2.State(register_response=rr)
            s.world_model.cycle = c
            try:
                call(s, timeout=10)
            except grpc.RpcError:
                break
            c += 1
    th = [threading.Thread(target=agent, args=(rr_p, stub.GetPlayerActions), daemon=True),
          threading.Thread(target=agent, args=(rr_t, stub.GetTrainerActions), daemon=True)]
    for t in th:
        t.start()
    dash = pb2.PlayerAction(dash=pb2.Dash(power=100, relative_direction=0))
    keep = pb2.TrainerAction(do_change_mode=pb2.DoChangeMode(game_mode_type=pb2.GameModeType.PlayOn, side=pb2.Side.LEFT))

    def step():                                  # soccer_2d_env.py:238-252
        paq.put(dash); taq.put(keep)
        psq.get(timeout=10); tsq.get(timeout=10)
    psq.get(timeout=10); tsq.get(timeout=10)     # the agents' first States
    for _ in range(100):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    stop.set()
    paq.put(dash); taq.put(keep)
    res = {'what': 'reference plumbing-only ceiling (server.serve + 4 mp.Queues + fake gRPC agents, no rcssserver/proxy)',
           'steps': steps, 'seconds': dt, 'steps_per_s': steps / dt, 'host_cores': os.cpu_count(),
           'where': 'build container (reference is not available on the GPU box)',
           'reference_paths': ['server.py:49-103,210-224', 'soccer_2d_env.py:62-65,238-252']}
    print(json.dumps(res))
    out = os.path.join(ROOT, 'profiles', 'r01', 'ref_plumbing_ceiling.json')
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    import signal
    try:
        os.killpg(proc.pid, signal.SIGTERM)  # exactly the group this probe started
    except ProcessLookupError:
        pass
    proc.join(timeout=5)
    os._exit(0)


if __name__ == '__main__':
    main()
