! racgpu_host.f90 -- Fortran host for the batched GPU chemistry solve.
!
!   racgpu_host <configure.dat> <cells.txt> <out_prefix> [nlocal_iter [restart.bin|- [ndev [hc.txt]]]]
!
!   nlocal_iter  (default 1) a_disk_iter_params%nlocal_iter of the reference's &iteration_configure: > 1 runs calc_this_cell's
!                local-iteration loop (retries from t_final with looser tolerance policies, reference src/disk.f90:1651-1791)
!                through racgpu_calc_cells
!   restart.bin  a chemical_data_iter_NNNN.bin of a previous run (reference src/data_dump.f90:88-162, read back as
!                back_cells_chemical_data_load does): the cells start from its abundances instead of the initial-abundance file
!                ('-': none)
!   ndev         (default 0) >= 1: the cells are dealt over ndev GPUs of this node by ONE process (racgpu_multi_calc_cells:
!                one host thread per device, one RCCL all-gather of the results); 0: the single-device entry points
!   hc.txt       per-cell heating/cooling records (RACGPU_NHC numbers per row, include/racgpu.h RACGPU_H_*): the gas temperature
!                co-evolves with the chemistry in every cell with en_gain_tot > 0 (chemsol_params%evolT, reference
!                src/disk.f90:2066-2073); the switches come from &heating_cooling_configure of <configure.dat> if it has one, the
!                enthalpy file from chemsol_params%filename_species_enthalpy, the ion look-up tables from
!                heating_cooling_config%dir_transition_rates, the Neufeld tables from <chem_files_dir>/neufeld_cooling_tables.dat
!
! Plays the part of rac-2d's cell sweep (do_chemical_stuff -> calc_this_cell, reference src/disk.f90:864-938,
! 1629-1801) for a table of frozen per-cell input records: reads the reference's own `&chemistry_configure`
! namelist, loads the network and initial abundances through the C ABI, sets every cell's initial condition
! (y <- y0, Grain0 <- ratioDust2HnucNum; src/disk.f90:2055-2066), solves all cells in one GPU call and writes
!   <out_prefix>.bin : direct-access records in the layout of the reference's chemical_data_iter_NNNN.bin
!                      (record i = cell i = abundances(nSpecies), col_den_toStar(10), col_den_toISM(10), all f64;
!                      reference src/data_dump.f90:88-162; the 20 column densities belong to the caller: zeros)
!   <out_prefix>.dat : the reference's iter_NNNN.dat (write_header + disk_save_results_write, src/disk.f90:2745-3073): '!' + 6 integer and
!                      142 real columns + the abundances, one row per cell; columns of subsystems outside this path are zero
!   <out_prefix>.counters : per cell NST, local iterations used, NERR (no counterpart in the reference's file)
!   with flag_chem_evol_save = .true. in the namelist, per cell i also
!   <out_prefix>_cellNNNNNN_<chem_evol_save_filename> : the time series chem_evol_solve writes while it integrates
!                      (reference src/chemistry.f90:404-413, 476-478): header '! Time', species names, 'Tgas' in A14,
!                      then one row (t, y(1:NEQ)) in ES14.4E4 per record 2..n_record_real
program racgpu_host
  use, intrinsic :: iso_c_binding
  use racgpu
  implicit none
  character(len=512) :: f_conf, f_cells, prefix, path, f_restart, argbuf, f_hc
  integer :: nlocal_iter, ndev
  type(c_ptr) :: multi
  type(racgpu_hc_config_t) :: hcc
  real(c_double), allocatable, target :: hc(:, :)
  logical :: evolT
  real(c_double), allocatable, target :: cell_out(:, :)
  real(c_double), allocatable :: col20(:)
  character(kind=c_char), dimension(32) :: nbuf
  character(len=12), allocatable :: names(:)
  type(c_ptr) :: net
  type(racgpu_params_t) :: p
  integer(c_int32_t) :: nS, nR, nnzJ, nzl, nzu
  integer :: fu, ios, ncell, i, k, rc, reclen, n_record, nrr, i_gH, i_H
  real(c_double), allocatable :: vrow(:)
  include 'iter_columns.inc'
  real(c_double), allocatable, target :: record(:, :, :), touts(:, :)
  type(c_ptr) :: prec, ptouts
  character(len=16) :: tag
  real(c_double), allocatable, target :: cells(:, :), y(:, :), y0(:), t_final(:)
  integer(c_int32_t), allocatable, target :: quality(:)
  integer(c_int64_t), allocatable, target :: stats(:, :)
  real(c_double) :: row(RACGPU_NPAR), zeros20(20)
  character(len=64) :: fmt

  if (command_argument_count() < 3) then
    write(*, '(A)') 'usage: racgpu_host <configure.dat> <cells.txt> <out_prefix>'
    stop 2
  end if
  call get_command_argument(1, f_conf)
  call get_command_argument(2, f_cells)
  call get_command_argument(3, prefix)
  nlocal_iter = 1
  f_restart = ''
  if (command_argument_count() >= 4) then
    call get_command_argument(4, argbuf)
    read(argbuf, *) nlocal_iter
  end if
  if (command_argument_count() >= 5) call get_command_argument(5, f_restart)
  if (trim(f_restart) == '-') f_restart = ''
  ndev = 0
  if (command_argument_count() >= 6) then
    call get_command_argument(6, argbuf)
    read(argbuf, *) ndev
  end if
  f_hc = ''
  if (command_argument_count() >= 7) call get_command_argument(7, f_hc)
  evolT = len_trim(f_hc) > 0

  open(newunit=fu, file=trim(f_conf), status='old', action='read')
  call chemistry_configure_read(fu, ios)
  close(fu)
  if (ios /= 0) then
    write(*, '(A, I6)') 'cannot read &chemistry_configure, iostat = ', ios
    stop 1
  end if
  call chemsol_to_c(p)
  if (evolT) then ! the reference reads the namelists in a fixed order (src/configure.f90:27-38); a file without this one keeps the defaults
    open(newunit=fu, file=trim(f_conf), status='old', action='read')
    call heating_cooling_configure_read(fu, ios)
    close(fu)
    call heating_cooling_to_c(hcc, 0.01D0, .true.)
  end if

  if (racgpu_device_count() < 1) then
    write(*, '(A)') 'racgpu_host: no HIP device visible (the racgpu path has no CPU fallback)'
    stop 1
  end if

  path = trim(chemsol_params%chem_files_dir) // trim(chemsol_params%filename_chemical_network)
  net = racgpu_network_load(c_string(trim(path)))
  if (.not. c_associated(net)) then
    write(*, '(A)') 'racgpu_network_load: ' // trim(racgpu_error_string())
    stop 1
  end if
  rc = racgpu_network_dims(net, nS, nR, nnzJ, nzl, nzu)
  write(*, '(A, I6, A, I6, A, I7, A, I7)') 'Number of species ', nS, '  reactions ', nR, '  nnz(J) ', nnzJ, '  nnz(LU) ', nzl + nzu + nS
  allocate(names(nS), y0(nS))
  do i = 1, nS
    rc = racgpu_species_name(net, int(i, c_int32_t), nbuf, 32_c_int32_t)
    names(i) = ''
    do k = 1, 12
      if (nbuf(k) == c_null_char) exit
      names(i)(k:k) = nbuf(k)
    end do
  end do
  path = trim(chemsol_params%chem_files_dir) // trim(chemsol_params%filename_initial_abundances)
  rc = racgpu_load_initial_abundances(net, c_string(trim(path)), y0)
  if (rc /= 0) then
    write(*, '(A)') 'racgpu_load_initial_abundances: ' // trim(racgpu_error_string())
    stop 1
  end if

  ! cell table: RACGPU_NPAR numbers per row
  ncell = 0
  open(newunit=fu, file=trim(f_cells), status='old', action='read')
  do
    read(fu, *, iostat=ios) row
    if (ios /= 0) exit
    ncell = ncell + 1
  end do
  rewind(fu)
  allocate(cells(RACGPU_NPAR, ncell), y(nS, ncell), t_final(ncell), quality(ncell), stats(RACGPU_NSTAT, ncell))
  do i = 1, ncell
    read(fu, *) cells(:, i)
  end do
  close(fu)

  if (evolT) then
    allocate(hc(RACGPU_NHC, ncell))
    open(newunit=fu, file=trim(f_hc), status='old', action='read')
    do i = 1, ncell
      read(fu, *) hc(:, i)
    end do
    close(fu)
    rc = racgpu_heating_cooling_load(net, hcc, &
           c_string(trim(chemsol_params%chem_files_dir) // trim(chemsol_params%filename_species_enthalpy)), &
           c_string(trim(chemsol_params%chem_files_dir) // 'neufeld_cooling_tables.dat'), &
           c_string(trim(heating_cooling_config%dir_transition_rates) // trim(heating_cooling_config%filename_NII)), &
           c_string(trim(heating_cooling_config%dir_transition_rates) // trim(heating_cooling_config%filename_SiII)), &
           c_string(trim(heating_cooling_config%dir_transition_rates) // trim(heating_cooling_config%filename_FeII)))
    if (rc /= 0) then
      write(*, '(A)') 'racgpu_heating_cooling_load: ' // trim(racgpu_error_string())
      stop 1
    end if
  end if
  rc = racgpu_init_abundances(net, y0, cells, int(ncell, c_int64_t), y)
  allocate(cell_out(RACGPU_NOUT, ncell), col20(20))
  cell_out = 0D0
  if (len_trim(f_restart) > 0) then ! use_backup_chemical_data: record i = cell i = abundances(nSpecies), 20 column densities
    inquire(iolength=reclen) y(:, 1), col20
    open(newunit=fu, file=trim(f_restart), access='direct', form='unformatted', recl=reclen, status='old', action='read')
    do i = 1, ncell
      read(fu, rec=i) y(:, i), col20
    end do
    close(fu)
    write(*, '(A, A)') 'Abundances taken from ', trim(f_restart)
  end if
  prec = c_null_ptr; ptouts = c_null_ptr
  n_record = racgpu_n_record(p, 0D0, p%t_max)
  if (chemsol_params%flag_chem_evol_save) then ! chemsol_stor%record / %touts of every cell (1.2 MB per cell)
    allocate(record(nS + 1, n_record, ncell), touts(n_record, ncell))
    prec = c_loc(record); ptouts = c_loc(touts)
  end if
  if (ndev >= 1) then
    ! one process, ndev GPUs: the cells dealt over the devices, one RCCL all-gather of the results (racgpu_multi_calc_cells)
    if (evolT .or. chemsol_params%flag_chem_evol_save) then
      write(*, '(A)') 'racgpu_host: the multi-GPU entry point runs the fixed-T local-iteration loop (no hc.txt, no flag_chem_evol_save)'
      stop 1
    end if
    multi = racgpu_multi_create(c_string(trim(chemsol_params%chem_files_dir) // trim(chemsol_params%filename_chemical_network)), &
                                int(ndev, c_int), c_null_ptr)
    if (.not. c_associated(multi)) then
      write(*, '(A)') 'racgpu_multi_create: ' // trim(racgpu_multi_error_string())
      stop 1
    end if
    rc = racgpu_multi_calc_cells(multi, p, int(nlocal_iter, c_int32_t), int(ncell, c_int64_t), c_loc(cells), c_loc(y), c_loc(t_final), &
                                 c_loc(quality), c_loc(stats), c_loc(cell_out), c_null_ptr)
    if (rc /= 0) then
      write(*, '(A)') 'racgpu_multi_calc_cells: ' // trim(racgpu_multi_error_string())
      stop 1
    end if
    call racgpu_multi_destroy(multi)
    write(*, '(A, I3, A)') 'Cells dealt over ', ndev, ' device(s), results gathered with one RCCL all-gather'
  else if (evolT) then
    if (nlocal_iter > 1) then
      write(*, '(A)') 'racgpu_host: with hc.txt (T evolving) nlocal_iter must be 1'
      stop 1
    end if
    rc = racgpu_evolT_solve_batch(net, p, int(ncell, c_int64_t), c_loc(cells), c_loc(hc), c_loc(y), c_null_ptr, c_null_ptr, c_loc(t_final), &
                                  c_loc(quality), c_loc(stats), prec, ptouts, c_loc(cell_out), 0_c_int, RACGPU_MEM_HOST)
  else if (nlocal_iter > 1) then
    if (chemsol_params%flag_chem_evol_save) then
      write(*, '(A)') 'racgpu_host: flag_chem_evol_save needs nlocal_iter = 1 (the reference overwrites the file in every local iteration)'
      stop 1
    end if
    rc = racgpu_calc_cells(net, p, int(nlocal_iter, c_int32_t), int(ncell, c_int64_t), c_loc(cells), c_loc(y), c_loc(t_final), &
                           c_loc(quality), c_loc(stats), c_loc(cell_out), RACGPU_MEM_HOST)
  else
    rc = racgpu_evol_solve_batch(net, p, int(ncell, c_int64_t), c_loc(cells), c_loc(y), c_null_ptr, c_null_ptr, c_loc(t_final), &
                                 c_loc(quality), c_loc(stats), prec, ptouts, c_loc(cell_out), 0_c_int, RACGPU_MEM_HOST)
  end if
  if (rc /= 0) then
    write(*, '(A)') 'racgpu solve: ' // trim(racgpu_error_string())
    stop 1
  end if
  if (ndev < 1) write(*, '(A, I8, A, F10.2, A, I12)') 'Solved ', ncell, ' cells; kernel ', racgpu_last_kernel_ms(net), ' ms; total steps ', sum(stats(1, :))

  zeros20 = 0D0
  inquire(iolength=reclen) y(:, 1), zeros20
  open(newunit=fu, file=trim(prefix) // '.bin', access='direct', form='unformatted', recl=reclen, status='replace')
  do i = 1, ncell
    write(fu, rec=i) y(:, i), zeros20
  end do
  close(fu)
  ! <out_prefix>.dat: the reference's iter_NNNN.dat (write_header + disk_save_results_write, src/disk.f90:2745-3073): "!" + 6 integer and
  ! 142 real columns + the abundances; the columns that belong to subsystems outside this path (geometry, photon counters, masses, fluxes)
  ! are written as the reference writes them for a cell those subsystems have not touched: zero
  open(newunit=fu, file=trim(prefix) // '.dat', status='replace')
  write(fmt, '("(A1, A4, A5, ", I4, "A14)")') 4 + n_iter_real + nS
  write(fu, fmt) '!', 'cvg', 'qual', adjustr('cr_count      '), adjustr('abc_dus       '), adjustr('scc_HI        '), adjustr('abc_wat       '), &
                 (adjustr(iter_real_names(i) // '      '), i = 1, n_iter_real), ('  ' // names(i), i = 1, nS)
  write(fmt, '("(2I5, 4I14, ", I4, "ES14.5E3)")') n_iter_real + nS
  allocate(vrow(n_iter_real))
  i_gH = 0; i_H = 0
  do i = 1, nS
    if (trim(names(i)) .eq. 'gH') i_gH = i
    if (trim(names(i)) .eq. 'H') i_H = i
  end do
  do i = 1, ncell
    vrow = 0D0
    vrow(icol('t_final')) = t_final(i);  vrow(icol('n_gas')) = cells(3, i);   vrow(icol('Tgas')) = cell_out(4, i)
    vrow(icol('Tdust')) = cells(2, i);   vrow(icol('ndust_t')) = cells(6, i); vrow(icol('sigd_av')) = cells(5, i)
    vrow(icol('d2gnum')) = cells(7, i);  vrow(icol('Av_ISM')) = cells(13, i); vrow(icol('Av_Star')) = cells(14, i)
    vrow(icol('UV_G0_I')) = cells(15, i); vrow(icol('UV_G0_S')) = cells(16, i)
    vrow(icol('LyANF0')) = cells(19, i); vrow(icol('LyAG0_a')) = cells(19, i) / 6D7   ! G0_Lya_atten (src/disk.f90:1880)
    vrow(icol('zeta_X')) = cells(11, i); vrow(icol('Ncol_I')) = cells(12, i)
    vrow(icol('f_H2_I')) = cells(20, i); vrow(icol('f_CO_I')) = cells(21, i); vrow(icol('f_H2O_I')) = cells(22, i); vrow(icol('f_OH_I')) = cells(23, i)
    vrow(icol('f_H2_S')) = cells(24, i); vrow(icol('f_CO_S')) = cells(25, i); vrow(icol('f_H2O_S')) = cells(26, i); vrow(icol('f_OH_S')) = cells(27, i)
    vrow(icol('nsit_gr')) = cells(8, i); vrow(icol('nmol_gr')) = cell_out(2, i)
    if (i_gH .gt. 0) then   ! get_H2_form_rate (src/disk.f90:4302-4315)
      vrow(icol('R_H2_fo')) = cell_out(1, i) * y(i_gH, i) * y(i_gH, i) * cells(3, i)
    else if (i_H .gt. 0) then
      vrow(icol('R_H2_fo')) = cell_out(1, i) * y(i_H, i) * cells(3, i)
    end if
    if (allocated(hc)) then
      do k = 1, 4
        write(tag, '(I1)') k
        vrow(icol('Tdust' // tag(1:1))) = hc(20 + k, i); vrow(icol('ndust_' // tag(1:1))) = hc(16 + k, i)
        vrow(icol('sigdus_' // tag(1:1))) = hc(12 + k, i); vrow(icol('egain_d' // tag(1:1))) = hc(24 + k, i)
      end do
      vrow(icol('egain_d')) = hc(1, i); vrow(icol('deplet')) = hc(10, i); vrow(icol('Ncol_S')) = hc(2, i)
      vrow(icol('w_Kep')) = hc(5, i);   vrow(icol('dv_turb')) = hc(6, i); vrow(icol('l_coher')) = hc(7, i)
    end if
    write(fu, fmt) 0, quality(i), 0, 0, 0, 0, vrow, y(:, i)
  end do
  close(fu)
  ! <out_prefix>.counters: what the engine adds per cell (no counterpart in the reference's file)
  open(newunit=fu, file=trim(prefix) // '.counters', status='replace')
  write(fu, '(A)') '!       NST    local_iter          NERR'
  do i = 1, ncell
    write(fu, '(3I14)') int(stats(1, i)), int(stats(18, i)), int(stats(5, i))
  end do
  close(fu)
  if (chemsol_params%flag_chem_evol_save) then
    do i = 1, ncell
      write(tag, '("_cell", I6.6, "_")') i
      open(newunit=fu, file=trim(prefix) // trim(tag) // trim(chemsol_params%chem_evol_save_filename), status='replace')
      write(fmt, '("(", I4, "A14)")') nS + 2
      write(fu, fmt) '! Time        ', names(1:nS), '   Tgas       '
      write(fmt, '("(", I4, "ES14.4E4)")') nS + 2
      nrr = int(stats(6, i)) ! RACGPU_S_NREC_REAL: chemsol_params%n_record_real
      do k = 2, min(nrr, n_record)
        write(fu, fmt) touts(k, i), record(:, k, i)
      end do
      close(fu)
    end do
  end if
  call racgpu_network_destroy(net)
contains
  integer function icol(name)
    character(len=*), intent(in) :: name
    integer :: k
    icol = 0
    do k = 1, n_iter_real
      if (trim(iter_real_names(k)) .eq. trim(name)) then
        icol = k
        return
      end if
    end do
    write(*, '(2A)') 'racgpu_host: unknown iter_NNNN.dat column ', name
    stop 1
  end function icol
end program racgpu_host
