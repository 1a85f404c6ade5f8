"""ctypes binding of librl4rs_hip.so (C ABI declared in include/rl4rs_hip.h).

There is NO CPU fallback: if the shared object is missing, or no HIP device is visible when a device
handle is created, this module raises.  Build the library with ``python -m rl4rs_amd.build`` or
``__graft_entry__.build()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'librl4rs_hip.so')

OK = 0


class Rl4rsHipError(RuntimeError):
    pass


class EnvCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'batch_size', 'max_steps', 'action_size', 'action_emb_size', 'page_items', 'item_dim',
        'user_dense_dim', 'user_cat_dim', 'maxlen', 'dense_feature_num', 'category_feature_num',
        'log_steps', 'is_seq', 'violation_zeroes_reward')]


class DienCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'maxlen', 'emb_size', 'hidden_units', 'dense_feature_num', 'category_feature_num',
        'category_hash_size', 'seq_num', 'class_num', 'max_rows', 'max_slots', 'scorer_mode')] + [('kernel_opts', C.c_uint32)]


# include/rl4rs_hip.h RL4RS_DIEN_OPT_*: kernel-path selection of one scorer handle (config['scorer_kernels'])
DIEN_OPTS = {'augru_h16': 1 << 0, 'augru_rows32': 1 << 1, 'augru_rows64': 1 << 2, 'din_v1': 1 << 3, 'no_din16': 1 << 4,
             'no_gru16': 1 << 5, 'no_gemm16': 1 << 6, 'no_cat16': 1 << 7, 'no_dense_chain': 1 << 8, 'no_head_tables': 1 << 9,
             'no_head_fused': 1 << 10, 'cat_v1': 1 << 11, 'no_cat_group': 1 << 12, 'dense_fork': 1 << 13, 'no_gru_pad': 1 << 14}
POLICY_OPTS = {'tile': 0, 'ppo_fused': 1, 'ppo_rows': 2, 'resident_wgs': 3, 'ppo_std': 4}          # RL4RS_POLICY_OPT_*
ENV_OPTS = {'rows_variant': 0}                                                       # RL4RS_ENV_OPT_*


_FP = C.POINTER(C.c_float)
_FP4 = _FP * 4


class DienWeights(C.Structure):
    _fields_ = [
        ('cat_emb', _FP),
        ('dense_w1', _FP), ('dense_b1', _FP), ('dense_w2', _FP), ('dense_b2', _FP),
        ('seq_emb', _FP),
        ('gru_gate_w', _FP4), ('gru_gate_b', _FP4), ('gru_cand_w', _FP4), ('gru_cand_b', _FP4),
        ('att_w1', _FP4), ('att_b1', _FP4), ('att_w2', _FP4), ('att_b2', _FP4),
        ('att_w3', _FP4), ('att_b3', _FP4),
        ('augru_gate_w', _FP4), ('augru_gate_b', _FP4), ('augru_cand_w', _FP4), ('augru_cand_b', _FP4),
        ('obs_w', _FP), ('obs_b', _FP), ('out_w', _FP), ('out_b', _FP),
    ]


class SimnetCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'algo', 'maxlen', 'emb_size', 'hidden_units', 'dense_feature_num', 'category_feature_num',
        'category_hash_size', 'seq_num', 'class_num', 'max_rows', 'max_slots')]


class SimnetWeights(C.Structure):
    _fields_ = [
        ('cat_emb', _FP), ('seq_emb', _FP),
        ('dense_w1', _FP), ('dense_b1', _FP), ('dense_w2', _FP), ('dense_b2', _FP),
        ('fc_w', _FP), ('fc_b', _FP), ('obs_w', _FP), ('obs_b', _FP), ('out_w', _FP), ('out_b', _FP),
        ('cat_gru_kernel', _FP), ('cat_gru_recurrent', _FP), ('cat_gru_bias', _FP),
        ('seq_gru_kernel', _FP4), ('seq_gru_recurrent', _FP4), ('seq_gru_bias', _FP4),
    ]


class QNetCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('obs_dim', 'action_size', 'mask_size', 'emb_size', 'hidden1', 'hidden2', 'n_layers',
                                         'max_rows')]


class AmlpCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('obs_dim', 'act_dim', 'hidden1', 'hidden2', 'out_dim', 'head_act', 'max_rows', 'max_grad_rows')]


class BcqStep(C.Structure):
    """rl4rs_bcq_step (include/rl4rs_hip.h)"""
    _fields_ = ([(n, C.c_void_p) for n in ('imit_enc', 'imit_dec', 'policy', 'policy_targ', 'q1', 'q2', 'q1_targ', 'q2_targ')] +
                [(n, C.c_int32) for n in ('B', 'n', 'E', 'L')] +
                [(n, C.c_float) for n in ('beta', 'action_flexibility', 'lam', 'gamma', 'tau', 'imitator_lr', 'critic_lr', 'actor_lr')] +
                [(n, C.c_int32) for n in ('do_rl', 'do_actor', 'nograd_h16', 'h16_min_rows')] +
                [(n, C.c_void_p) for n in ('obs_dev', 'act_dev', 'rew_dev', 'nxt_dev', 'ter_dev', 'noise_dev', 'workspace_dev', 'metrics_dev')])


class CqlStep(C.Structure):
    """rl4rs_cql_step (include/rl4rs_hip.h)"""
    _fields_ = ([(n, C.c_void_p) for n in ('policy', 'q1', 'q2', 'q1_targ', 'q2_targ')] +
                [(n, C.c_int32) for n in ('B', 'n', 'A')] +
                [(n, C.c_float) for n in ('gamma', 'tau', 'actor_lr', 'critic_lr', 'temp_lr', 'alpha_lr', 'alpha_threshold', 'conservative_weight')] +
                [(n, C.c_int32) for n in ('nograd_h16', 'h16_min_rows')] +
                [(n, C.c_int64) for n in ('temp_step', 'alpha_step')] +
                [(n, C.c_void_p) for n in ('log_temp_dev', 'log_alpha_dev', 'obs_dev', 'act_dev', 'rew_dev', 'nxt_dev', 'ter_dev', 'normal_dev',
                                           'uniform_dev', 'workspace_dev', 'metrics_dev')])


class RawPolicyCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'maxlen', 'emb_size', 'hidden_units', 'dense_feature_num', 'category_feature_num', 'category_hash_size',
        'seq_num', 'action_size', 'max_rows')]


class RawPolicyWeights(C.Structure):
    _fields_ = [(n, _FP) for n in ('cat_emb', 'seq_emb', 'dense_w1', 'dense_b1', 'dense_w2', 'dense_b2', 'ctx_w', 'ctx_b',
                                   'out_w', 'out_b', 'value_w', 'value_b')]


class StepRecord(C.Structure):
    """rl4rs_step_record: byte offsets of the parts of a transition record (-1 = absent)."""
    _fields_ = [(n, C.c_int64) for n in ('status', 'reward', 'done', 'chosen', 'obs', 'obs_d3rl', 'mask_i64', 'mask_bits', 'click_p',
                                         'offline_action', 'host_bytes', 'total_bytes')] + [('obs_dim', C.c_int32), ('d3rl_cols', C.c_int32)]


STEP_WANT = {'mask_i64': 1, 'mask_bits': 2, 'd3rl_obs': 4, 'click_p': 8, 'offline_action': 16}      # RL4RS_STEP_WANT_*

_lib = None

# name -> (restype, argtypes); every symbol include/rl4rs_hip.h declares
_P = C.c_void_p
_I = C.c_int
_I32 = C.c_int32
_I64 = C.c_int64
SIGNATURES = {
    'rl4rs_last_error': (C.c_char_p, []),
    'rl4rs_abi_version': (_I, []),
    'rl4rs_device_count': (_I, []),
    'rl4rs_copy_d2d': (_I, [_P, _P, _I64, _P]),
    'rl4rs_env_create': (_I, [C.POINTER(EnvCfg), C.POINTER(_P)]),
    'rl4rs_env_destroy': (_I, [_P]),
    'rl4rs_env_set_catalog': (_I, [_P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_env_load_batch': (_I, [_P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_env_load_lines': (_I, [_P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, _I32, _P, _P]),
    'rl4rs_parse_records': (_I, [C.c_char_p, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, C.POINTER(_I32)]),
    'rl4rs_crc32c': (C.c_uint32, [_P, _I64, C.c_uint32]),
    'rl4rs_env_reset': (_I, [_P, _P]),
    'rl4rs_env_act_discrete': (_I, [_P, _P, _P]),
    'rl4rs_env_act_conti': (_I, [_P, _P, _I, _P, _P]),
    'rl4rs_knn': (_I, [_P, _I, _I32, _P, _I32, _I32, _P, _P, _P]),
    'rl4rs_env_build_complete': (_I, [_P, _P]),
    'rl4rs_env_build_complete_rows': (_I, [_P, _I32, _P]),
    'rl4rs_env_reward_split': (_I, [_P, _P, _P, _P, _P]),
    'rl4rs_env_complete_rows': (_I, [_P]),
    'rl4rs_env_is_reward_step': (_I, [_P]),
    'rl4rs_env_cur_steps': (_I, [_P]),
    'rl4rs_env_reward': (_I, [_P, _P, _P, _P]),
    'rl4rs_env_violation': (_I, [_P, _P, _P]),
    'rl4rs_env_obs_mask': (_I, [_P, _P, _I, _P]),
    'rl4rs_env_offline_action': (_I, [_P, _P, _P, _P]),
    'rl4rs_env_offline_reward': (_I, [_P, _P, _P]),
    'rl4rs_env_predict_with_mask': (_I, [_P, _I32, _P, _P, _I32, _P, _P, _P]),
    'rl4rs_env_buffer': (_I, [_P, _I, C.POINTER(_P), C.POINTER(_I64)]),
    'rl4rs_dien_create': (_I, [C.POINTER(DienCfg), C.POINTER(DienWeights), _P, C.POINTER(_P)]),
    'rl4rs_dien_destroy': (_I, [_P]),
    'rl4rs_dien_scorer_mode': (_I, [_P, C.POINTER(_I32)]),
    'rl4rs_dien_status': (_I, [_P, C.POINTER(_I32), _P]),
    'rl4rs_dien_encode': (_I, [_P, _I32, _P, _I32, _I32, _P]),
    'rl4rs_dien_forward': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_dien_head_prob': (_I, [_P, _I32, _P, _P, _P]),
    'rl4rs_dien_buffer': (_I, [_P, _I, C.POINTER(_P), C.POINTER(_I64)]),
    'rl4rs_simnet_create': (_I, [C.POINTER(SimnetCfg), C.POINTER(SimnetWeights), _P, C.POINTER(_P)]),
    'rl4rs_simnet_destroy': (_I, [_P]),
    'rl4rs_simnet_obs_dim': (_I, [_P, C.POINTER(_I32)]),
    'rl4rs_simnet_encode': (_I, [_P, _I32, _P, _I32, _I32, _P]),
    'rl4rs_simnet_forward': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_simnet_head_prob': (_I, [_P, _I32, _P, _P, _P]),
    'rl4rs_dien_set_profiling': (_I, [_P, _I]),
    'rl4rs_dien_kernel_count': (_I, []),
    'rl4rs_dien_kernel_name': (C.c_char_p, [_I]),
    'rl4rs_dien_kernel_label': (_I, [_P, _I, C.c_char_p, _I32]),
    'rl4rs_dien_profile_read': (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(_I64)]),
    'rl4rs_dien_profile_reset': (_I, [_P]),
    'rl4rs_policy_param_count': (_I, [_I32, _I32, _I32]),
    'rl4rs_policy_create': (_I, [_I32, _I32, _I32, _I32, _P, _P, C.POINTER(_P)]),
    'rl4rs_policy_destroy': (_I, [_P]),
    'rl4rs_policy_params': (_I, [_P, C.POINTER(_P), C.POINTER(_I32)]),
    'rl4rs_policy_act': (_I, [_P, _I32, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_policy_evaluate': (_I, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_policy_loss_grad': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, _P, _P, _P]),
    'rl4rs_policy_adam_step': (_I, [_P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'rl4rs_simtrain_create': (_I, [C.POINTER(SimnetCfg), C.POINTER(SimnetWeights), _I32, _P, C.POINTER(_P)]),
    'rl4rs_simtrain_destroy': (_I, [_P]),
    'rl4rs_simtrain_params': (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    'rl4rs_simtrain_masks': (_I, [_P, C.POINTER(_P), C.POINTER(_P)]),
    'rl4rs_simtrain_grad': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_float, C.c_uint32, C.c_uint32, _P, _P]),
    'rl4rs_simtrain_step': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_uint32, C.c_uint32, _P, _P]),
    'rl4rs_recur_train_set_rows': (_I, [_I32]),
    'rl4rs_dientrain_create': (_I, [C.POINTER(DienCfg), C.POINTER(DienWeights), _I32, _P, C.POINTER(_P)]),
    'rl4rs_dientrain_destroy': (_I, [_P]),
    'rl4rs_dientrain_set_fork': (_I, [_I32]),
    'rl4rs_dientrain_params': (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    'rl4rs_dientrain_masks': (_I, [_P, C.POINTER(_P), C.POINTER(_P)]),
    'rl4rs_dientrain_grad': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_float, C.c_uint32, C.c_uint32, _P, _P]),
    'rl4rs_dientrain_step': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_uint32, C.c_uint32, _P, _P]),
    'rl4rs_rawpolicy_create': (_I, [C.POINTER(RawPolicyCfg), C.POINTER(RawPolicyWeights), _P, C.POINTER(_P)]),
    'rl4rs_rawpolicy_destroy': (_I, [_P]),
    'rl4rs_rawpolicy_act': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_rawpolicy_evaluate': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_rawtrain_create': (_I, [C.POINTER(RawPolicyCfg), C.POINTER(RawPolicyWeights), _P, C.POINTER(_P)]),
    'rl4rs_rawtrain_destroy': (_I, [_P]),
    'rl4rs_rawtrain_params': (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    'rl4rs_rawtrain_act': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_rawtrain_evaluate': (_I, [_P, _I32, _P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_rawtrain_loss_grad': (_I, [_P, _I32, _I32, _P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_float, _P, _P]),
    'rl4rs_rawtrain_adam_step': (_I, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'rl4rs_policy_ppo_epoch': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P] + [C.c_float] * 10 + [_P, _P, _P]),
    'rl4rs_dien_set_row_order': (_I, [_P, _P, _I32]),
    'rl4rs_dien_set_augru_rows': (_I, [_P, _I32]),
    'rl4rs_dien_status_word': (_I, [_P, C.POINTER(_P)]),
    'rl4rs_stepper_record_layout': (_I, [_P, C.c_uint32, _I32, C.POINTER(StepRecord)]),
    'rl4rs_env_step_record': (_I, [_P, _P, _I32, C.c_uint32, _P, _P]),
    'rl4rs_env_step_record_host': (_I, [_P, _P, _I32, C.c_uint32, _P, _P, _P]),
    'rl4rs_env_observe_record_host': (_I, [_P, _I32, C.c_uint32, _P, _P, _P]),
    'rl4rs_set_host_mirror': (_I, [_I32]),
    'rl4rs_env_set_option': (_I, [_P, _I32, _I32]),
    'rl4rs_policy_set_option': (_I, [_P, _I32, _I32]),
    'rl4rs_env_get_cfg': (_I, [_P, _P]),
    'rl4rs_env_attach_scorer': (_I, [_P, _P, _P, _I32, _P]),
    'rl4rs_env_attach_simnet': (_I, [_P, _P, _P, _I32, _P]),
    'rl4rs_stepper_destroy': (_I, [_P]),
    'rl4rs_env_step_discrete': (_I, [_P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_env_step_conti': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'rl4rs_policy_ppo_minibatch_grad': (_I, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P] + [C.c_float] * 5 + [_P, _P, _P]),
    'rl4rs_policy_status': (_I, [_P, _P, _P]),
    'rl4rs_policy_status_words': (_I, [_P, _P]),
    'rl4rs_policy_adam_state': (_I, [_P, _P, _P, _P]),
    'rl4rs_policy_set_adam_step': (_I, [_P, C.c_int64]),
    'rl4rs_qnet_create': (_I, [_P, _FP, _P, _P, _P, _P]),
    'rl4rs_qnet_destroy': (_I, [_P]),
    'rl4rs_qnet_params': (_I, [_P, _P, _P, _P]),
    'rl4rs_qnet_copy_params': (_I, [_P, _P, _P]),
    'rl4rs_qnet_adam_state': (_I, [_P, _P, _P, _P]),
    'rl4rs_qnet_set_adam_step': (_I, [_P, _I64]),
    'rl4rs_qnet_status': (_I, [_P, C.POINTER(_I32), _P]),
    'rl4rs_qnet_forward': (_I, [_P, _I32, _P, _P, _P]),
    'rl4rs_qnet_backward': (_I, [_P, _I32, _P, _P, _P]),
    'rl4rs_qnet_adam_step': (_I, [_P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'rl4rs_q_best_action': (_I, [_I32, _I32, _P, _P, C.c_float, _P, _P]),
    'rl4rs_qloss_imitation': (_I, [_P, _I32, _P, _P, C.c_float, _P, _P, _P, _P]),
    'rl4rs_qloss_dqn': (_I, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    'rl4rs_amlp_create': (_I, [_P, _FP, _P, _P]),
    'rl4rs_amlp_destroy': (_I, [_P]),
    'rl4rs_amlp_params': (_I, [_P, _P, _P, _P]),
    'rl4rs_amlp_copy_params': (_I, [_P, _P, _P]),
    'rl4rs_amlp_adam_state': (_I, [_P, _P, _P, _P]),
    'rl4rs_amlp_set_adam_step': (_I, [_P, _I64]),
    'rl4rs_amlp_soft_update': (_I, [_P, _P, C.c_float, _P]),
    'rl4rs_amlp_forward': (_I, [_P, _I32, _I32, _P, _P, _P, _P]),
    'rl4rs_amlp_h16_ok': (_I, [_P]),
    'rl4rs_pack_h16_selftest': (_I, [_P, _I64, _I32, _I32, _P, _P]),
    'rl4rs_amlp_forward_h16': (_I, [_P, _I32, _I32, _P, _P, _P, _P]),
    'rl4rs_amlp_backward': (_I, [_P, _I32, _I32, _P, _P, _P, _P, _I32, _P]),
    'rl4rs_amlp_adam_step': (_I, [_P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'rl4rs_amlp_adam_multi': (_I, [_I32, C.POINTER(_P), C.POINTER(C.c_float), C.POINTER(_I32), C.POINTER(_P), C.c_float, C.c_float, C.c_float,
                                   C.c_float, _P]),
    'rl4rs_amlp_set_fused': (_I, [_I32]),
    'rl4rs_cql_workspace_floats': (_I64, [_I32, _I32, _I32]),
    'rl4rs_cql_update': (_I, [C.POINTER(CqlStep), _P]),
    'rl4rs_bcq_workspace_floats': (_I64, [_I32, _I32, _I32, _I32]),
    'rl4rs_bcq_update': (_I, [C.POINTER(BcqStep), _P]),
    'rl4rs_amlp_forward_multi': (_I, [_I32, C.POINTER(_P), _I32, _P, _P, C.POINTER(_P), _P]),
    'rl4rs_amlp_backward_multi': (_I, [_I32, C.POINTER(_P), _I32, _P, _P, C.POINTER(_P), C.POINTER(_P), _I32, _P]),
    'rl4rs_cvae_sample': (_I, [_I32, _I32, _P, _P, C.c_float, C.c_float, _P, _P]),
    'rl4rs_cvae_loss': (_I, [_I32, _I32, _I32, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P]),
    'rl4rs_cvae_encoder_grad': (_I, [_I32, _I32, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P]),
    'rl4rs_residual_action': (_I, [_I32, _I32, _P, _P, C.c_float, _P, _P]),
    'rl4rs_residual_grad': (_I, [_I32, _I32, _P, _P, C.c_float, _P, _P, _P]),
    'rl4rs_bcq_target': (_I, [_I32, _I32, _P, _P, C.c_float, _P, _P, C.c_float, _P, _P, _P]),
    'rl4rs_pick_rows': (_I, [_I32, _I32, _I32, _P, _P, _P, _P]),
    'rl4rs_critic_mse': (_I, [_I32, _P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_squashed_sample': (_I, [_I32, _I32, _I32, _P, _P, C.c_float, C.c_float, _I32, _I32, _P, _P, _P]),
    'rl4rs_sac_actor_grad': (_I, [_I32, _I32, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P]),
    'rl4rs_twin_min': (_I, [_I32, _P, _P, _P, _P, _P, _P]),
    'rl4rs_cql_critic_loss': (_I, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'rl4rs_gemm_f32_packed': (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I, _P]),
    'rl4rs_gemm_h16_packed': (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I, _P]),
    'rl4rs_gemm_f32': (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I, _P]),
}


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get('RL4RS_LIB') or LIB_PATH          # RL4RS_LIB: A/B measurements of differently built libraries
    if not os.path.exists(lib_path):
        raise Rl4rsHipError(
            "librl4rs_hip.so not found at %s: build it with `python -m rl4rs_amd.build` "
            "(hipcc --offload-arch=gfx950). rl4rs_amd has no CPU fallback." % lib_path)
    # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime this process uses, so import it
    # before dlopen()ing the library (two HIP runtimes in one process cannot see each other's devices).
    import torch  # noqa: F401
    lib = C.CDLL(lib_path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.rl4rs_abi_version() != 1:
        raise Rl4rsHipError("librl4rs_hip.so ABI version %d, expected 1" % lib.rl4rs_abi_version())
    if os.environ.get('RL4RS_HOST_MIRROR', '') in ('0', '1'):     # A/B runs of unmodified scripts (the library never reads the environment)
        lib.rl4rs_set_host_mirror(int(os.environ['RL4RS_HOST_MIRROR']))
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().rl4rs_last_error()
        raise Rl4rsHipError("librl4rs_hip error %d: %s" % (rc, msg.decode() if msg else '?'))
    return rc


def require_device():
    """Raise unless a HIP device is visible (called before any device handle is created)."""
    n = load().rl4rs_device_count()
    if n <= 0:
        raise Rl4rsHipError("no MI355X / HIP device visible (rl4rs_device_count=%d): "
                            "rl4rs_amd runs the env step on the GPU only" % n)
    return n
