"""`rodent_amd.envs`: mirror of the `brax.envs` registry the launcher uses
[REF brax_rodent_run_ppo.py:57,82-90]."""
from .base import Contact, PipelineEnv, PipelineState, State
from .rodent import Rodent
from . import wrappers

_envs = {"rodent": Rodent}


def register_environment(env_name: str, env_class):
    _envs[env_name] = env_class


def get_environment(env_name: str, **kwargs):
    return _envs[env_name](**kwargs)
