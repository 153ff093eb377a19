"""The two helpers train.py uses from rl/networks/network_utils.py (:45-49 update_linear_schedule)."""


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
