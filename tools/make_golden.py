"""Transcribe the known-answer DATA stored in the reference's notebook outputs into small JSON fixtures
under tests/golden/ (SURVEY.md Appendix C).  Run where /root/reference exists; the fixtures are committed.
Only stored numeric outputs are copied (no notebook source)."""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NUM = r"-?\d+\.\d*(?:e[+-]?\d+)?"


def cell_outputs(nb_path):
    nb = json.load(open(nb_path))
    for c in nb["cells"]:
        for o in c.get("outputs", []):
            if "data" in o and "text/plain" in o["data"]:
                yield "".join(o["data"]["text/plain"])
            elif "text" in o:
                yield "".join(o["text"])


def main():
    os.makedirs(OUT, exist_ok=True)
    env_step = list(cell_outputs(os.path.join(REF, "Env_step.ipynb")))
    # (1) qpos after reset(PRNGKey(0)) of the notebook's env: qpos0 + U(-.01,.01) from split(PRNGKey(0),3)[1]
    qpos = next(t for t in env_step if t.startswith("Array([ 4.0483385e-02"))
    qpos = [float(x) for x in re.findall(NUM, qpos.split("dtype")[0])]
    # (2) qfrc_actuator at that state
    qfrc = next(t for t in env_step if "-3.71473469e-02" in t)
    qfrc = [float(x) for x in re.findall(NUM, qfrc.split("dtype")[0])]
    assert len(qpos) == 74 and len(qfrc) == 73
    json.dump({"source": "Env_step.ipynb stored outputs (cells 31 and 51)", "reset_qpos": qpos, "qfrc_actuator": qfrc},
              open(os.path.join(OUT, "env_step_reset.json"), "w"), indent=0)
    # (2b) the WHOLE observation of that reset state, printed one value per line by the notebook (cell 8, 1260 numbers):
    # qpos(74) | qvel(73) | cinert[1:](65 x 10) | cvel[1:](65 x 6) | qfrc_actuator(73) -- the reference's own mjx.forward output
    obs = next(t for t in env_step if t.startswith("0.040483385\n"))
    obs = [float(x) for x in obs.split()]
    assert len(obs) == 1260 and obs[:3] == [0.040483385, -0.008967142, 0.075515956]
    json.dump({"source": "Env_step.ipynb stored output of `for i in rodent_state.obs: print(i)` after reset(PRNGKey(0)); "
                         "layout qpos(74) qvel(73) cinert[1:](650) cvel[1:](390) qfrc_actuator(73)", "obs": obs},
              open(os.path.join(OUT, "env_step_obs.json"), "w"), indent=0)
    # (3) brax sys.link_names order and the contact struct of mjcf.ipynb
    mj = list(cell_outputs(os.path.join(REF, "mjcf.ipynb")))
    names = next(t for t in mj if t.startswith("['torso'"))
    names = re.findall(r"'([^']+)'", names)
    contact = next(t for t in mj if t.startswith("Contact(dist=Array"))

    def arr(field, nxt):
        seg = contact.split(field + "=Array(", 1)[1].split(nxt, 1)[0]
        return [float(x) for x in re.findall(NUM, seg.split("dtype")[0])]

    def ints(field, nxt):
        seg = contact.split(field + "=", 1)[1].split(nxt, 1)[0]
        return [int(x) for x in re.findall(r"-?\d+", re.sub(r"dtype=int32", "", seg))]

    golden = {"source": "mjcf.ipynb stored outputs (cells 7 and 20), model rodent_optimized.xml of 2024-03",
              "link_names": names,
              "contact_dist": arr("dist", "pos="),
              "contact_pos": arr("pos", "frame="),
              "contact_frame": arr("frame", "includemargin="),
              "contact_friction": arr("friction", "solref="),
              "contact_solref": arr("solref", "solreffriction="),
              "contact_solimp": arr("solimp", "geom1="),
              "contact_geom1": ints("geom1", "geom2="),
              "contact_geom2": ints("geom2", "link_idx="),
              "contact_link_idx2": ints("link_idx", "elasticity=")[42:]}
    json.dump(golden, open(os.path.join(OUT, "mjcf_contact_struct.json"), "w"), indent=0)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in golden.items()})


if __name__ == "__main__":
    main()
