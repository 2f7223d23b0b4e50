"""TEST INFRASTRUCTURE, opt-in (tests/test_gym_env.py, RDV_RUN_REFERENCE_SOURCE=1, build container only): runs three functions of the
reference FROM THEIR SOURCE TEXT on ``RendezvousEnv`` in this child process.  The reference's code is given ``np`` and a no-op table
printer, nothing else (no ``os``, no ``pickle``: its ``args.save`` branch is never taken); nothing of it is stored in the repo."""
import os
import sys
import textwrap
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]

import numpy as np  # noqa: E402

import test_gym_env as T  # noqa: E402


def cut(path, start_marker, end_marker, include_end=False, dedent=False):
    src = open(path).read()
    a = src.index(start_marker)
    b = src.index(end_marker, a) + (len(end_marker) if include_end else 0)
    text = textwrap.dedent(src[a:b]) if dedent else src[a:b]
    return compile(text, path, "exec")


def main(ref):
    scope = {"np": np}
    exec(cut(os.path.join(ref, "monte_carlo.py"), "def evaluate(model, env, initial_state):", 'if __name__ == "__main__":'), scope)
    T.check_mc_evaluate(scope["evaluate"])
    scope = {"np": np, "print_state": lambda env: None}     # print_state: a table printer (utils/environment_utils.py:100)
    exec(cut(os.path.join(ref, "save_new_trajectory.py"), "def evaluate(model, env, args):", "def get_args():"), scope)
    evaluate = scope["evaluate"]
    T.check_record_trajectory(lambda model, env: evaluate(model, env, types.SimpleNamespace(save=False)))
    scope = {"np": np}
    exec(cut(os.path.join(ref, "custom", "custom_callbacks.py"), "    def evaluate_policy(self):", "        return output", include_end=True, dedent=True), scope)
    method = scope["evaluate_policy"]
    T.check_callback_evaluation(lambda model, env, n: method(types.SimpleNamespace(model=model, env=env, n_evals=n)))
    print("3 reference functions ran unchanged on RendezvousEnv")


if __name__ == "__main__":
    main(sys.argv[1])
