! ref_driver.f90 -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
!
! A driver of our own that links against the UNMODIFIED reference objects (everything except its
! main program) and calls only the reference's routines, following the single-cell fixed-T recipe
! of SURVEY.md section 8(c) (reference: src/disk.f90:1566-1581 setup sequence, src/disk.f90:2014-2100
! initial condition, src/chemistry.f90:391 chem_evol_solve).  It exists to (1) validate the C
! restatement in oracle/ and (2) generate the golden vectors committed under tests/golden/.
! It is never part of the product path and never travels as source to the GPU box.
!
! Input : a namelist file (argument 1) + a whitespace-separated cell table (one cell per row,
!         28 columns = the racgpu cell record, see include/racgpu.h).
! Output: plain-text vectors, one value per line, %ES25.17E3.
program ref_driver
  use chemistry
  use heating_cooling
  use disk, only: a_disk, write_header, disk_save_results_write
  use data_struct, only: type_cell
  use trivials, only: double2str
  implicit none
  external chem_ode_f, chem_ode_jac
  character(len=256) :: nml_file, chem_dir, network, initial, out_dir, cell_file
  integer :: ncell, mxstep, steps_reset, dump_jac, dump_record_every, solve
  double precision :: rtol, atol, dt_first_step, ratio_tstep, t_max
  logical :: h2_moeq, special_gH_mobi
  ! nlocal_iter > 1: the local-iteration loop of calc_this_cell (src/disk.f90:1651-1791) around chem_evol_solve;
  ! tol_j: j of chem_set_solver_flags_alt for a single pass; y_override: file of rows "cell species value" applied to
  ! the initial condition (to provoke the sanity exits of src/chemistry.f90:520-530)
  integer :: nlocal_iter, tol_j, dump_analysis
  character(len=9) :: strtmp
  double precision, allocatable :: flux(:)
  integer :: ie, isp, fA
  character(len=256) :: y_override
  ! evolT = 1: gas temperature co-evolution (chemsol_params%evolT, src/disk.f90:2069-2073).  hc_file: one row of NHC numbers per
  ! cell = the fields of the cell record that only the heating/cooling terms read (layout: include/racgpu.h, RACGPU_H_*);
  ! enthalpy: the species-enthalpy file (chemical heating); transitions_dir: where the ion-cooling tables N+/Si+/Fe+_LUT.bin lie.
  ! The heating/cooling switches are the README template's (README.md:135-156).
  integer :: evolT, may_switch_T, dump_iter_file
  type(type_cell), pointer :: cc
  integer :: fI
  character(len=256) :: hc_file, enthalpy, transitions_dir
  integer, parameter :: NHC = 28
  double precision :: hpar(NHC), Tdot
  integer :: fH
  namelist /ref_run/ chem_dir, network, initial, out_dir, cell_file, ncell, &
    rtol, atol, dt_first_step, ratio_tstep, t_max, mxstep, steps_reset, h2_moeq, &
    dump_jac, dump_record_every, solve, nlocal_iter, tol_j, y_override, special_gH_mobi, dump_analysis, &
    evolT, may_switch_T, hc_file, enthalpy, transitions_dir, dump_iter_file
  double precision, allocatable :: abund(:)
  double precision :: t_final, t_end, dt0, tmp, ov_val
  integer :: jj, isav, qual_cell, ov_cell, ov_spe, fO, ios
  integer, parameter :: NPAR = 28
  double precision :: cpar(NPAR)
  double precision, allocatable :: ydot(:), pdj(:), dummy(:)
  integer :: ic, i, j, k, fU, fC, NEQ, nS, c0, c1, crate
  character(len=300) :: fname
  double precision :: tdummy

  call get_command_argument(1, nml_file)
  chem_dir = './'; network = ''; initial = ''; out_dir = './'; cell_file = ''
  ncell = 1; rtol = 1D-4; atol = 1D-30; dt_first_step = 1D-8; ratio_tstep = 1.1D0
  t_max = 1D6; mxstep = 6000; steps_reset = 50; h2_moeq = .false.
  dump_jac = 1; dump_record_every = 0; solve = 1
  nlocal_iter = 1; tol_j = 1; y_override = ''; special_gH_mobi = .false.; dump_analysis = 0
  evolT = 0; may_switch_T = 1; hc_file = ''; enthalpy = ''; transitions_dir = './transitions/'; dump_iter_file = 0
  open(newunit=fU, file=trim(nml_file), status='old', action='read')
  read(fU, nml=ref_run)
  close(fU)

  chemsol_params%chem_files_dir = chem_dir
  chemsol_params%filename_chemical_network = network
  chemsol_params%filename_initial_abundances = initial
  chemsol_params%filename_species_enthalpy = enthalpy
  chemsol_params%RTOL = rtol
  chemsol_params%ATOL = atol
  chemsol_params%t0 = 0D0
  chemsol_params%t_max = t_max
  chemsol_params%dt_first_step = dt_first_step
  chemsol_params%ratio_tstep = ratio_tstep
  chemsol_params%max_runtime_allowed = 1.0E9
  chemsol_params%mxstep_per_interval = mxstep
  chemsol_params%steps_reset_solver = steps_reset
  chemsol_params%H2_form_use_moeq = h2_moeq
  chemsol_params%flag_chem_evol_save = .false.
  chemsol_params%evol_dust_size = .false.
  chemsol_params%use_special_gH_mobi = special_gH_mobi
  open(newunit=fU, file=trim(out_dir)//'/ref_log.txt', status='replace', action='write')
  chemsol_params%fU_log = fU

  ! The reference's own setup sequence (src/disk.f90:1566-1581, minus enthalpies).
  call chem_read_reactions()
  call chem_load_reactions()
  call chem_parse_reactions()
  call chem_get_dupli_reactions()
  call chem_get_idx_for_special_species()
  call chem_make_sparse_structure
  call chem_prepare_solver_storage
  call chem_evol_solve_prepare_run_once
  call chem_load_initial_abundances
  if (evolT .ne. 0) then
    ! src/disk.f90:1573-1575 (enthalpies -> reaction heats), :1643 (hc_params => chem_params), heating_cooling_prepare
    call chem_load_species_enthalpies
    call chem_get_reaction_heat
    heating_cooling_config%dir_transition_rates = transitions_dir
    heating_cooling_config%use_analytical_CII_OI = .true.
    heating_cooling_config%IonCoolingWithLut = .true.
    heating_cooling_config%filename_NII = 'N+_LUT.bin'
    heating_cooling_config%filename_SiII = 'Si+_LUT.bin'
    heating_cooling_config%filename_FeII = 'Fe+_LUT.bin'
    heating_cooling_config%solve_method = 2
    heating_cooling_config%use_mygasgraincooling = .true.
    heating_cooling_config%use_chemicalheatingcooling = .true.
    heating_cooling_config%use_Xray_heating = .true.
    heating_cooling_config%heating_Xray_en = 0D0
    heating_cooling_config%heating_eff_chem = 0.3D0
    heating_cooling_config%heating_eff_H2form = 0.5D0
    heating_cooling_config%heating_eff_phd_H2 = 1D0
    heating_cooling_config%heating_eff_phd_H2O = 0.5D0
    heating_cooling_config%heating_eff_phd_OH = 0.5D0
    heating_cooling_config%cooling_gg_coeff = 1D0
    call heating_cooling_prepare
    hc_params => chem_params
    a_disk%allow_gas_dust_en_exch = .false.
    a_disk%Tdust_iter_tandem = .false.
    a_disk%base_alpha = 0.01D0
    open(newunit=fH, file=trim(hc_file), status='old', action='read')
    open(newunit=fC, file=trim(out_dir)//'/heat.txt', status='replace')
    write(fC, '(I8)') chem_net%nReacWithHeat
    do i = 1, chem_net%nReacWithHeat
      write(fC, '(I8, X, ES25.17E3)') chem_net%iReacWithHeat(i), chem_net%heat(i)
    end do
    close(fC)
  end if

  nS = chem_species%nSpecies
  NEQ = chemsol_params%NEQ
  allocate(ydot(NEQ), pdj(NEQ), dummy(1), abund(nS))

  ! ---- network-level dumps -------------------------------------------------
  open(newunit=fC, file=trim(out_dir)//'/species.txt', status='replace')
  do i = 1, nS
    write(fC, '(A)') trim(chem_species%names(i))
  end do
  close(fC)
  open(newunit=fC, file=trim(out_dir)//'/network.txt', status='replace')
  write(fC, '(3I8)') nS, chem_net%nReactions, chemsol_params%NNZ
  do i = 1, chem_net%nReactions
    write(fC, '(11I6)') chem_net%reac(:, i), chem_net%prod(:, i), chem_net%n_reac(i), &
      chem_net%n_prod(i), chem_net%itype(i), chem_net%dupli(i)%nItem
  end do
  close(fC)
  open(newunit=fC, file=trim(out_dir)//'/species_attr.txt', status='replace')
  do i = 1, nS
    write(fC, '(3ES25.17E3, I8, I4)') chem_species%mass_num(i), chem_species%vib_freq(i), &
      chem_species%Edesorb(i), chem_species%idx_gasgrain_counterpart(i), chem_species%elements(1, i)
  end do
  close(fC)
  open(newunit=fC, file=trim(out_dir)//'/pattern.txt', status='replace')
  do i = 1, NEQ + 1
    write(fC, '(I8)') chemsol_stor%IWORK(30 + i)
  end do
  do i = 1, chemsol_params%NNZ
    write(fC, '(I8)') chemsol_stor%IWORK(31 + NEQ + i)
  end do
  close(fC)
  open(newunit=fC, file=trim(out_dir)//'/y0.txt', status='replace')
  do i = 1, nS
    write(fC, '(ES25.17E3)') chemsol_stor%y0(i)
  end do
  close(fC)

  if (dump_iter_file .ne. 0) then
    ! the reference's own iter_NNNN.dat writer (write_header, disk_save_results_write: src/disk.f90:2745-3073) on ONE cell whose every
    ! printed field holds a number that says which field it is: field k of the row prints k + k/1000 (integers: k), abundance i: i * 1e-3
    allocate(cc); allocate(cc%par); allocate(cc%h_c_rates)
    allocate(cc%abundances(nS_probe()), cc%col_den_toISM(chem_idx_some_spe%nItem), cc%col_den_toStar(chem_idx_some_spe%nItem))
    call fill_probe_cell(cc)
    open(newunit=fI, file=trim(out_dir)//'/iter_probe.dat', status='replace')
    call write_header(fI)
    call disk_save_results_write(fI, cc)
    close(fI)
  end if

  open(newunit=fU, file=trim(cell_file), status='old', action='read')
  do ic = 1, ncell
    read(fU, *) cpar
    write(chemsol_params%fU_log, '(A, I8)') '# cell ', ic  ! the error handler's "!Error: ISTATE" lines that follow belong to this cell
    chem_params%Tgas                      = cpar(1)
    chem_params%Tdust                     = cpar(2)
    chem_params%n_gas                     = cpar(3)
    chem_params%GrainRadius_CGS           = cpar(4)
    chem_params%sigdust_ave               = cpar(5)
    chem_params%ndust_tot                 = cpar(6)
    chem_params%ratioDust2HnucNum         = cpar(7)
    chem_params%SitesPerGrain             = cpar(8)
    chem_params%omega_albedo              = cpar(9)
    chem_params%zeta_cosmicray_H2         = cpar(10)
    chem_params%zeta_Xray_H2              = cpar(11)
    chem_params%Ncol_toISM                = cpar(12)
    chem_params%Av_toISM                  = cpar(13)
    chem_params%Av_toStar                 = cpar(14)
    chem_params%G0_UV_toISM               = cpar(15)
    chem_params%G0_UV_toStar              = cpar(16)
    chem_params%G0_UV_H2phd               = cpar(17)
    chem_params%G0_UV_toStar_photoDesorb  = cpar(18)
    chem_params%phflux_Lya                = cpar(19)
    chem_params%f_selfshielding_toISM_H2  = cpar(20)
    chem_params%f_selfshielding_toISM_CO  = cpar(21)
    chem_params%f_selfshielding_toISM_H2O = cpar(22)
    chem_params%f_selfshielding_toISM_OH  = cpar(23)
    chem_params%f_selfshielding_toStar_H2 = cpar(24)
    chem_params%f_selfshielding_toStar_CO = cpar(25)
    chem_params%f_selfshielding_toStar_H2O= cpar(26)
    chem_params%f_selfshielding_toStar_OH = cpar(27)
    if (evolT .ne. 0) then
      read(fH, *) hpar
      chem_params%en_gain_tot     = hpar(1)
      chem_params%Ncol_toStar     = hpar(2)
      chem_params%PAH_abundance   = hpar(3)
      chem_params%MeanMolWeight   = hpar(4)
      chem_params%omega_Kepler    = hpar(5)
      chem_params%velo_width_turb = hpar(6)
      chem_params%coherent_length = hpar(7)
      chem_params%Neufeld_G       = hpar(8)
      chem_params%Neufeld_dv_dz   = hpar(9)
      chem_params%dust_depletion  = hpar(10)
      chem_params%volume          = hpar(11)
      chem_params%ndustcompo      = int(hpar(12))
      chem_params%sig_dusts       = hpar(13:16)
      chem_params%n_dusts         = hpar(17:20)
      chem_params%Tdusts          = hpar(21:24)
      chem_params%en_gains        = hpar(25:28)
      chem_params%X_gH            = 0D0
    end if

    ! src/disk.f90:2055-2075 (set_initial_condition_4solver), fixed-T branch.
    chemsol_stor%y(1:nS) = chemsol_stor%y0(1:nS)
    if (chem_idx_some_spe%i_Grain0 .ne. 0) then
      chemsol_stor%y(chem_idx_some_spe%i_Grain0) = chem_params%ratioDust2HnucNum
    end if
    chemsol_stor%y(nS + 1) = chem_params%Tgas
    if (len_trim(y_override) .gt. 0) then
      open(newunit=fO, file=trim(y_override), status='old', action='read')
      do
        read(fO, *, iostat=ios) ov_cell, ov_spe, ov_val
        if (ios .ne. 0) exit
        if (ov_cell .eq. ic) chemsol_stor%y(ov_spe) = ov_val
      end do
      close(fO)
    end if
    chemsol_params%evolT = .false.
    chemsol_params%maySwitchT = .false.
    if (evolT .ne. 0) then ! src/disk.f90:2069-2073
      chemsol_params%evolT = .true.
      chemsol_params%maySwitchT = may_switch_T .ne. 0
      if (chem_params%en_gain_tot .le. 0D0) chemsol_params%evolT = .false.
    end if
    chemsol_params%t0 = 0D0
    chemsol_params%dt_first_step = dt_first_step
    chemsol_params%t_max = t_max
    if (cpar(28) .gt. 0D0) chemsol_params%t_max = cpar(28)
    call chem_evol_solve_prepare_ongoing
    ! src/disk.f90:1671-1686 order: flags, rates, solve.
    call chem_set_solver_flags_alt(tol_j)
    call chem_cal_rates

    write(fname, '(A, "/cell_", I4.4, ".txt")') trim(out_dir), ic
    open(newunit=fC, file=trim(fname), status='replace')
    write(fC, '(A, I8)') '# rates ', chem_net%nReactions
    do i = 1, chem_net%nReactions
      write(fC, '(ES25.17E3)') chem_net%rates(i)
    end do
    write(fC, '(A, I8)') '# rtol ', NEQ
    do i = 1, NEQ
      write(fC, '(ES25.17E3)') chemsol_stor%RTOLs(i)
    end do
    write(fC, '(A, I8)') '# atol ', NEQ
    do i = 1, NEQ
      write(fC, '(ES25.17E3)') chemsol_stor%ATOLs(i)
    end do
    tdummy = 0D0
    call chem_ode_f(NEQ, tdummy, chemsol_stor%y, ydot)
    write(fC, '(A, I8)') '# ydot0 ', NEQ
    do i = 1, NEQ
      write(fC, '(ES25.17E3)') ydot(i)
    end do
    if (evolT .ne. 0) call dump_hc(fC, 'hc0 ')
    if (dump_jac .ne. 0) then
      write(fC, '(A, I8)') '# jac0 ', chemsol_params%NNZ
      do j = 1, NEQ
        pdj = 0D0
        call chem_ode_jac(NEQ, tdummy, chemsol_stor%y, j, dummy, dummy, pdj)
        do k = chemsol_stor%IWORK(30 + j), chemsol_stor%IWORK(31 + j) - 1
          write(fC, '(ES25.17E3)') pdj(chemsol_stor%IWORK(31 + NEQ + k))
        end do
      end do
    end if

    if ((solve .ne. 0) .and. (nlocal_iter .gt. 1)) then
      ! calc_this_cell's loop (src/disk.f90:1651-1791) with the continue rule of set_initial_condition_4solver_continue
      ! (src/disk.f90:2103-2146); every routine called is the reference's own.  One '# iter' section per local iteration:
      ! j, t0, dt_first_step, n_record, touts(n_record_real), quality, NERR, isav, t_final handed back, n_mol_on_grain,
      ! proceeds(1/0); then '# yiter' = the abundances the cell holds after the iteration.
      dt0 = dt_first_step
      t_final = 0D0
      abund = chemsol_stor%y(1:nS)
      qual_cell = 0
      do jj = 1, nlocal_iter
        if (jj .gt. 1) then
          chemsol_stor%y(1:nS) = abund
          chemsol_stor%y(nS + 1) = chem_params%Tgas
          call rectify_abundances(NEQ, chemsol_stor%y)
          chemsol_params%t0 = t_final
          chemsol_params%dt_first_step = max(dt0, chemsol_params%t0 * 1D-3)
          call chem_evol_solve_prepare_ongoing
          call chem_set_solver_flags_alt(jj)
          call chem_cal_rates
        end if
        call chem_evol_solve
        t_end = chemsol_stor%touts(chemsol_params%n_record_real)
        write(fC, '(A, I8, I8)') '# iter ', 11, jj
        write(fC, '(ES25.17E3)') dble(jj)
        write(fC, '(ES25.17E3)') chemsol_params%t0
        write(fC, '(ES25.17E3)') chemsol_params%dt_first_step
        write(fC, '(ES25.17E3)') dble(chemsol_params%n_record)
        write(fC, '(ES25.17E3)') t_end
        if ((jj .gt. 1) .and. (t_end .le. t_final)) then
          ! 'Local iteration does not proceed': nothing is taken over
          write(fC, '(ES25.17E3)') dble(qual_cell)
          write(fC, '(ES25.17E3)') dble(chemsol_params%NERR)
          write(fC, '(ES25.17E3)') 0D0
          write(fC, '(ES25.17E3)') t_final
          write(fC, '(ES25.17E3)') chem_params%n_mol_on_grain
          write(fC, '(ES25.17E3)') 0D0
          exit
        end if
        do isav = chemsol_params%n_record_real, 1, -1
          if ((.not. isnan(chemsol_stor%record(nS + 1, isav))) .and. &
              (.not. isnan(chemsol_stor%record(chem_idx_some_spe%i_H2, isav)))) exit
        end do
        qual_cell = chemsol_params%quality
        if (isav .gt. 1) then
          abund = chemsol_stor%record(1:nS, isav)
          t_final = chemsol_stor%touts(isav)
          tmp = get_ice_coverage(nS, abund)
        end if
        write(fC, '(ES25.17E3)') dble(qual_cell)
        write(fC, '(ES25.17E3)') dble(chemsol_params%NERR)
        write(fC, '(ES25.17E3)') dble(isav)
        write(fC, '(ES25.17E3)') t_final
        write(fC, '(ES25.17E3)') chem_params%n_mol_on_grain
        write(fC, '(ES25.17E3)') 1D0
        write(fC, '(A, I8, I8)') '# yiter ', nS, jj
        do i = 1, nS
          write(fC, '(ES25.17E3)') abund(i)
        end do
        if (isav .le. 1) exit
        if ((qual_cell .eq. 0) .or. (t_final .ge. 0.5D0 * chemsol_params%t_max)) exit
      end do
      write(fC, '(A, I8)') '# rh2form ', 1
      write(fC, '(ES25.17E3)') chem_params%R_H2_form_rate_coeff
    else if (solve .ne. 0) then
      call system_clock(c0, crate)
      call chem_evol_solve
      call system_clock(c1)
      write(fC, '(A, I8)') '# yend ', NEQ
      do i = 1, NEQ
        write(fC, '(ES25.17E3)') chemsol_stor%y(i)
      end do
      write(fC, '(A, I8)') '# scalars ', 4
      write(fC, '(ES25.17E3)') chemsol_stor%touts(chemsol_params%n_record_real)
      write(fC, '(ES25.17E3)') dble(chemsol_params%quality)
      write(fC, '(ES25.17E3)') dble(chemsol_params%NERR)
      write(fC, '(ES25.17E3)') dble(c1 - c0) / dble(crate)
      ! IWORK(11,12,13,21): NST NFE NJE NLU of the LAST solver segment (zeroed by every ISTATE=1)
      write(fC, '(A, I8)') '# workspace ', 4
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(17)) ! LENRW: RWORK length DLSODES actually needs
      write(fC, '(ES25.17E3)') dble(chemsol_params%LRW)     ! length the reference allocates (20 + 4 NNZ + 28 NEQ)
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(22)) ! LYH
      write(fC, '(ES25.17E3)') dble(chemsol_params%NNZ)
      write(fC, '(A, I8)') '# stats ', 9
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(11))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(12))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(13))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(21))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(19))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(25))
      write(fC, '(ES25.17E3)') dble(chemsol_stor%IWORK(26))
      write(fC, '(ES25.17E3)') dble(chemsol_params%n_record)
      write(fC, '(ES25.17E3)') dble(chemsol_params%n_record_real)
      write(fC, '(A, I8)') '# sideeffects ', 2
      tmp = get_ice_coverage(nS, chemsol_stor%y(1:nS))
      write(fC, '(ES25.17E3)') chem_params%R_H2_form_rate_coeff
      write(fC, '(ES25.17E3)') chem_params%n_mol_on_grain
      write(fC, '(A, I8)') '# touts ', chemsol_params%n_record
      do i = 1, chemsol_params%n_record
        write(fC, '(ES25.17E3)') chemsol_stor%touts(i)
      end do
      if (dump_record_every .gt. 0) then
        do k = 1, chemsol_params%n_record, dump_record_every
          write(fC, '(A, I8, I8)') '# record ', NEQ, k
          do i = 1, NEQ
            write(fC, '(ES25.17E3)') chemsol_stor%record(i, k)
          end do
        end do
      end if
      if (dump_analysis .ne. 0) then
        ! the reference's own analysis of the end state: chem_ode_f_alt, get_contribution_each, chem_elemental_residence
        ! (src/chemistry.f90:1593-1854) and the rows of save_chem_rates (src/disk.f90:3573-3589)
        allocate(flux(chem_net%nReactions))
        call chem_ode_f_alt(chem_net%nReactions, flux, NEQ, chemsol_stor%y)
        write(fC, '(A, I8)') '# flux ', chem_net%nReactions
        do i = 1, chem_net%nReactions
          write(fC, '(ES25.17E3)') flux(i)
        end do
        deallocate(flux)
        call get_species_produ_destr
        call get_contribution_each
        do k = 1, 10
          isp = chem_idx_some_spe%idx(k)
          write(fC, '(A, I8, I8)') '# produ ', 2 * min(chem_species%produ(isp)%nItem, 20), isp
          do j = 1, min(chem_species%produ(isp)%nItem, 20)
            write(fC, '(ES25.17E3)') dble(chem_species%produ(isp)%list(j))
            write(fC, '(ES25.17E3)') chem_species%produ(isp)%contri(j)
          end do
          write(fC, '(A, I8, I8)') '# destr ', 2 * min(chem_species%destr(isp)%nItem, 20), isp
          do j = 1, min(chem_species%destr(isp)%nItem, 20)
            write(fC, '(ES25.17E3)') dble(chem_species%destr(isp)%list(j))
            write(fC, '(ES25.17E3)') chem_species%destr(isp)%contri(j)
          end do
        end do
        call chem_elemental_residence
        do ie = 1, const_nElement
          write(fC, '(A, I8, I8)') '# eleres ', 3 * chem_ele_resi(ie)%n_nonzero, ie
          do j = 1, chem_ele_resi(ie)%n_nonzero
            write(fC, '(ES25.17E3)') dble(chem_ele_resi(ie)%iSpecies(j))
            write(fC, '(ES25.17E3)') chem_ele_resi(ie)%ele_frac(j)
            write(fC, '(ES25.17E3)') chem_ele_resi(ie)%ele_accu(j)
          end do
        end do
        write(fname, '(A, "/ratedump_", I4.4, ".txt")') trim(out_dir), ic
        open(newunit=fA, file=trim(fname), status='replace')
        do k = 1, chem_net%nReactions
          call double2str(strtmp, chem_net%ABC(3, k), 9, 1)
          write(fA, '(7(A12), ES9.2, F9.2, A9, 2I6, I3, X, A1, X, A2, ES16.6E3)') &
            chem_net%reac_names(:,k), chem_net%prod_names(:,k), chem_net%ABC(1:2,k), strtmp, &
            int(chem_net%T_range(:,k)), chem_net%itype(k), chem_net%reliability(k), chem_net%ctype(k), chem_net%rates(k)
        end do
        close(fA)
      end if
      ! RHS at the end state: second ydot pin, at a chemically evolved composition (with T evolving again if the T-freeze test had
      ! switched it off: the dump is of chem_ode_f's evolT branch)
      if ((evolT .ne. 0) .and. (chem_params%en_gain_tot .gt. 0D0)) then
        write(fC, '(A, I8)') '# evolTend ', 1  ! 1: T was still evolving at the end of the run, 0: the T-freeze test had switched it off
        if (chemsol_params%evolT) then
          write(fC, '(ES25.17E3)') 1D0
        else
          write(fC, '(ES25.17E3)') 0D0
        end if
        chemsol_params%evolT = .true.
      end if
      call chem_ode_f(NEQ, tdummy, chemsol_stor%y, ydot)
      write(fC, '(A, I8)') '# ydotend ', NEQ
      do i = 1, NEQ
        write(fC, '(ES25.17E3)') ydot(i)
      end do
      if (evolT .ne. 0) then
        call dump_hc(fC, 'hcend ')
        write(fC, '(A, I8)') '# Trecord ', chemsol_params%n_record
        do i = 1, chemsol_params%n_record
          write(fC, '(ES25.17E3)') chemsol_stor%record(nS + 1, i)
        end do
      end if
    end if
    close(fC)
  end do
  close(fU)
contains
  integer function nS_probe()
    nS_probe = chem_species%nSpecies
  end function nS_probe

  subroutine fill_probe_cell(c)
    ! field k of disk_save_results_write's list gets the value k + k/1000 (the four integer counters: k)
    type(type_cell), pointer, intent(inout) :: c
    integer :: i
    double precision :: v(148)
    do i = 1, 148
      v(i) = dble(i) + dble(i) * 1D-3
    end do
    c%converged = .true.; c%quality = 2
    c%par%ab_count_dust = 4; c%par%sc_count_HI = 5; c%par%ab_count_water = 6
    c%par%t_final = v(7); c%xmin = v(8); c%xmax = v(9); c%ymin = v(10); c%ymax = v(11)
    c%par%n_gas = v(12); c%par%Tgas = v(13); c%par%Tdust = v(14); c%par%Tdusts = v(15:18); c%par%n_dusts = v(19:22); c%par%ndust_tot = v(23)
    c%par%rho_dusts = v(24:27); c%par%sig_dusts = v(28:31); c%par%sigdust_ave = v(32); c%par%ratioDust2GasMass = v(33)
    c%par%ratioDust2HnucNum = v(34); c%par%dust_depletion = v(35); c%par%mgas_cell = v(36); c%par%mdust_tot = v(37)
    c%par%pressure_thermal = v(38); c%par%area_T = 1D0; c%par%gravity_acc_z = v(39)
    c%par%en_gain_tot = v(40); c%par%en_gain_abso_tot = v(41); c%par%en_exchange_tot = v(42)
    c%par%en_gains = (/v(43), v(45), v(47), v(49)/); c%par%en_exchange = (/v(44), v(46), v(48), v(50)/)
    c%par%flux_tot = v(51); c%par%flux_Xray = v(52); c%par%flux_UV = v(53) * phy_Habing_energy_flux_CGS; c%par%flux_Lya = v(54)
    c%par%flux_Vis = v(55); c%par%flux_NIR = v(56); c%par%flux_MIR = v(57); c%par%flux_FIR = v(58)
    c%par%dir_tot_r = v(59); c%par%dir_tot_z = v(60); c%par%aniso_tot = v(61); c%par%dir_Xray_r = v(62); c%par%dir_Xray_z = v(63); c%par%aniso_Xray = v(64)
    c%par%dir_UV_r = v(65); c%par%dir_UV_z = v(66); c%par%aniso_UV = v(67); c%par%dir_Lya_r = v(68); c%par%dir_Lya_z = v(69); c%par%aniso_Lya = v(70)
    c%par%dir_Vis_r = v(71); c%par%dir_Vis_z = v(72); c%par%aniso_Vis = v(73); c%par%dir_NIR_r = v(74); c%par%dir_NIR_z = v(75); c%par%aniso_NIR = v(76)
    c%par%dir_MIR_r = v(77); c%par%dir_MIR_z = v(78); c%par%aniso_MIR = v(79); c%par%dir_FIR_r = v(80); c%par%dir_FIR_z = v(81); c%par%aniso_FIR = v(82)
    c%par%Av_toISM = v(83); c%par%Av_toStar = v(84); c%par%G0_UV_toISM = v(85); c%par%G0_UV_toStar = v(86); c%par%G0_Lya_atten = v(87)
    c%par%phflux_Lya = v(88); c%par%zeta_Xray_H2 = v(89); c%par%Ncol_toISM = v(90); c%par%Ncol_toStar = v(91)
    c%col_den_toISM = 0D0; c%col_den_toStar = 0D0
    c%col_den_toISM(chem_idx_some_spe%iiH2) = v(92); c%col_den_toISM(chem_idx_some_spe%iiH2O) = v(93)
    c%col_den_toISM(chem_idx_some_spe%iiOH) = v(94); c%col_den_toISM(chem_idx_some_spe%iiCO) = v(95)
    c%col_den_toStar(chem_idx_some_spe%iiH2) = v(96); c%col_den_toStar(chem_idx_some_spe%iiH2O) = v(97)
    c%col_den_toStar(chem_idx_some_spe%iiOH) = v(98); c%col_den_toStar(chem_idx_some_spe%iiCO) = v(99)
    c%par%f_selfshielding_toISM_H2 = v(100); c%par%f_selfshielding_toISM_H2O = v(101); c%par%f_selfshielding_toISM_OH = v(102)
    c%par%f_selfshielding_toISM_CO = v(103); c%par%f_selfshielding_toStar_H2 = v(104); c%par%f_selfshielding_toStar_H2O = v(105)
    c%par%f_selfshielding_toStar_OH = v(106); c%par%f_selfshielding_toStar_CO = v(107); c%par%R_H2_form_rate = v(108)
    associate(r => c%h_c_rates)
      r%hc_net_rate = v(109); r%heating_photoelectric_small_grain_rate = v(110); r%heating_formation_H2_rate = v(111)
      r%heating_cosmic_ray_rate = v(112); r%heating_vibrational_H2_rate = v(113); r%heating_ionization_CI_rate = v(114)
      r%heating_photodissociation_H2_rate = v(115); r%heating_photodissociation_H2O_rate = v(116); r%heating_photodissociation_OH_rate = v(117)
      r%heating_Xray_Bethell_rate = v(118); r%heating_viscosity_rate = v(119); r%heating_chem = v(120)
      r%cooling_photoelectric_small_grain_rate = v(121); r%cooling_vibrational_H2_rate = v(122); r%cooling_gas_grain_collision_rate = v(123)
      r%cooling_OI_rate = v(124); r%cooling_CII_rate = v(125); r%cooling_NII_rate = v(126); r%cooling_SiII_rate = v(127); r%cooling_FeII_rate = v(128)
      r%cooling_OH_rot_rate = v(129); r%cooling_Neufeld_H2O_rate_rot = v(130); r%cooling_Neufeld_H2O_rate_vib = v(131)
      r%cooling_Neufeld_CO_rate_rot = v(132); r%cooling_Neufeld_CO_rate_vib = v(133); r%cooling_Neufeld_H2_rot_rate = v(134)
      r%cooling_LymanAlpha_rate = v(135); r%cooling_free_bound_rate = v(136); r%cooling_free_free_rate = v(137)
    end associate
    c%par%alpha_viscosity = v(138); c%par%ambipolar_f = v(139); c%par%ion_charge = v(140); c%par%velo_Kepler = v(141); c%par%omega_Kepler = v(142)
    c%par%velo_gradient = v(143); c%par%sound_speed = v(144); c%par%velo_width_turb = v(145); c%par%coherent_length = v(146)
    c%par%SitesPerGrain = v(147); c%par%n_mol_on_grain = v(148)
    do i = 1, chem_species%nSpecies
      c%abundances(i) = dble(i) * 1D-3
    end do
  end subroutine fill_probe_cell

  subroutine dump_hc(fC, tag)
    ! the 28 heating/cooling terms of the last chem_ode_f call (heating_minus_cooling, src/heating_cooling.f90:1204-1269), erg s-1 cm-3,
    ! in the order of type_heating_cooling_rates_list (src/data_struct.f90:489-520): net first
    integer, intent(in) :: fC
    character(len=*), intent(in) :: tag
    write(fC, '(A, A, I8)') '# ', tag, 29
    associate(r => heating_cooling_rates)
      write(fC, '(ES25.17E3)') r%hc_net_rate
      write(fC, '(ES25.17E3)') r%heating_photoelectric_small_grain_rate
      write(fC, '(ES25.17E3)') r%heating_formation_H2_rate
      write(fC, '(ES25.17E3)') r%heating_cosmic_ray_rate
      write(fC, '(ES25.17E3)') r%heating_vibrational_H2_rate
      write(fC, '(ES25.17E3)') r%heating_ionization_CI_rate
      write(fC, '(ES25.17E3)') r%heating_photodissociation_H2_rate
      write(fC, '(ES25.17E3)') r%heating_photodissociation_H2O_rate
      write(fC, '(ES25.17E3)') r%heating_photodissociation_OH_rate
      write(fC, '(ES25.17E3)') r%heating_Xray_Bethell_rate
      write(fC, '(ES25.17E3)') r%heating_viscosity_rate
      write(fC, '(ES25.17E3)') r%heating_chem
      write(fC, '(ES25.17E3)') r%cooling_photoelectric_small_grain_rate
      write(fC, '(ES25.17E3)') r%cooling_vibrational_H2_rate
      write(fC, '(ES25.17E3)') r%cooling_gas_grain_collision_rate
      write(fC, '(ES25.17E3)') r%cooling_OI_rate
      write(fC, '(ES25.17E3)') r%cooling_CII_rate
      write(fC, '(ES25.17E3)') r%cooling_Neufeld_H2O_rate_rot
      write(fC, '(ES25.17E3)') r%cooling_Neufeld_H2O_rate_vib
      write(fC, '(ES25.17E3)') r%cooling_Neufeld_CO_rate_rot
      write(fC, '(ES25.17E3)') r%cooling_Neufeld_CO_rate_vib
      write(fC, '(ES25.17E3)') r%cooling_Neufeld_H2_rot_rate
      write(fC, '(ES25.17E3)') r%cooling_LymanAlpha_rate
      write(fC, '(ES25.17E3)') r%cooling_free_bound_rate
      write(fC, '(ES25.17E3)') r%cooling_free_free_rate
      write(fC, '(ES25.17E3)') r%cooling_NII_rate
      write(fC, '(ES25.17E3)') r%cooling_SiII_rate
      write(fC, '(ES25.17E3)') r%cooling_FeII_rate
      write(fC, '(ES25.17E3)') r%cooling_OH_rot_rate
    end associate
  end subroutine dump_hc
end program ref_driver
